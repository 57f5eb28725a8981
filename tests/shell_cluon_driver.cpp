// Test driver of the cluon binding (tests/test_shell.py::test_cluon_binding_...): a lap of the synthetic track as ENCODED
// cluon Envelopes — every message serialised to the OD4 wire format and parsed back (cluon::serializeEnvelope /
// extractEnvelope) before it reaches the triggers — through csrc/gs_shell_cluon.hpp, in-process (no UDP).  What the shell
// publishes is encoded by the typed od4.send path into Envelopes as well, decoded again and printed, one line per
// message: "type senderStamp sampleTimeUs objectId v0 v1 v2" (%.17g), for the Python shell fed the same stream to match.
// Built by oracle/Makefile (ref_shell) next to the microservice binary; needs a GPU to run.
#include <cstdio>
#include <sstream>
#include <vector>

#include "../opendlv-logic-cfsd18-sensation-slam_amd/csrc/gs_shell_cluon.hpp"

extern "C" int gs_track_generate(int32_t, int32_t, double *, double *, double *, int32_t *, double *, int32_t *);
extern "C" int gs_track_obs_per_pose(void);

// stands where OD4Session stands: the same encode OD4Session::send does (reference src/cluon-complete-build.hpp:7809-7826), no socket
struct LoopbackOd4 {
    std::vector<std::string> wire;
    template <class T> void send(T &message, const cluon::data::TimeStamp &sampleTimeStamp, uint32_t senderStamp) {
        cluon::ToProtoVisitor enc; cluon::data::Envelope env;
        env.dataType(static_cast<int32_t>(message.ID())); message.accept(enc); env.serializedData(enc.encodedData());
        env.sent(cluon::time::now()); env.sampleTimeStamp(sampleTimeStamp); env.senderStamp(senderStamp);
        wire.push_back(cluon::serializeEnvelope(std::move(env)));
    }
};
template <class T> static cluon::data::Envelope roundtrip(T &msg, int64_t sample_us, uint32_t stamp) {
    LoopbackOd4 lo; lo.send(msg, cluon::time::fromMicroseconds(sample_us), stamp);
    std::stringstream ss(lo.wire[0]);
    auto r = cluon::extractEnvelope(ss);
    if (!r.first) { std::fprintf(stderr, "envelope did not survive the wire format\n"); std::exit(2); }
    return r.second;
}

int main(int argc, char **argv) {
    const int N = 120, M = 60, K = gs_track_obs_per_pose();
    std::vector<double> truth(3 * N), odom(3 * N), cxy(2 * M), obs((size_t)4 * K * N); std::vector<int32_t> ctype(M), ocone((size_t)K * N);
    if (gs_track_generate(N, M, truth.data(), odom.data(), cxy.data(), ctype.data(), obs.data(), ocone.data()) != 0) return 3;
    LoopbackOd4 out;
    ShellCluon shell(argc, argv, -1, [&out](const gs_shell_msg &o) { ShellCluon::sendWith(out, o); });
    if (shell.status() != GS_OK) { std::fprintf(stderr, "%s\n", gs_last_error()); return 1; }
    const double ref[2] = {57.70924648, 11.9462};
    int64_t now = 1000000;
    for (int n = 0; n < N + 8; ++n) { const int k = n < N ? n : n - N;
        now += 600000;
        const int64_t sample = 50000000 + 100000 * (int64_t)n;
        double latlon[2]; gs_wgs84_from_cartesian(ref, &odom[3 * (size_t)k], latlon);
        { opendlv::logic::sensation::Geolocation g; g.latitude(latlon[0]); g.longitude(latlon[1]); g.heading(static_cast<float>(odom[3 * (size_t)k + 2]));
          shell.onEnvelope(roundtrip(g, sample, 112), now);
          shell.onEnvelope(roundtrip(g, sample, 7), now); }                               // a foreign sender
        { opendlv::proxy::AngularVelocityReading w; w.angularVelocityZ(0.01f * static_cast<float>(n % 7));
          shell.onEnvelope(roundtrip(w, sample + 30000, 112), now); }
        for (int i = 0; i < K; ++i) { const double *o = &obs[(size_t)4 * ((size_t)k * K + i)];
            opendlv::logic::perception::ObjectDirection d; d.objectId(i); d.azimuthAngle(static_cast<float>(o[0])); d.zenithAngle(static_cast<float>(o[1]));
            opendlv::logic::perception::ObjectDistance r; r.objectId(i); r.distance(static_cast<float>(o[2]));
            opendlv::logic::perception::ObjectType t; t.objectId(i); t.type(static_cast<uint32_t>(o[3]));
            shell.onEnvelope(roundtrip(t, sample + 10000, 116), now + 100);
            shell.onEnvelope(roundtrip(d, sample + 10000, 116), now + 100);
            shell.onEnvelope(roundtrip(r, sample + 10000, 116), now + 100);
            shell.onEnvelope(roundtrip(r, sample + 10000, 3), now + 100); }              // a foreign sender's cone
        if (shell.poll(now + 100 + 20001) < 0) { std::fprintf(stderr, "%s\n", gs_last_error()); return 1; }
    }
    // decode what was published, exactly as a receiver of the OD4 session would
    for (auto &w : out.wire) { std::stringstream ss(w); auto r = cluon::extractEnvelope(ss); if (!r.first) return 2;
        cluon::data::Envelope env = r.second; const int32_t ty = env.dataType(); const uint32_t st = env.senderStamp();
        const long long us = cluon::time::toMicroseconds(env.sampleTimeStamp());
        if (ty == opendlv::logic::sensation::Geolocation::ID()) { auto m = cluon::extractMessage<opendlv::logic::sensation::Geolocation>(std::move(env));
            std::printf("%d %u %lld 0 %.17g %.17g %.17g\n", ty, st, us, m.latitude(), m.longitude(), (double)m.heading()); }
        else if (ty == opendlv::logic::perception::ObjectDirection::ID()) { auto m = cluon::extractMessage<opendlv::logic::perception::ObjectDirection>(std::move(env));
            std::printf("%d %u %lld %u %.17g %.17g 0\n", ty, st, us, m.objectId(), (double)m.azimuthAngle(), (double)m.zenithAngle()); }
        else if (ty == opendlv::logic::perception::ObjectDistance::ID()) { auto m = cluon::extractMessage<opendlv::logic::perception::ObjectDistance>(std::move(env));
            std::printf("%d %u %lld %u %.17g 0 0\n", ty, st, us, m.objectId(), (double)m.distance()); }
        else if (ty == opendlv::logic::perception::ObjectType::ID()) { auto m = cluon::extractMessage<opendlv::logic::perception::ObjectType>(std::move(env));
            std::printf("%d %u %lld %u %.17g 0 0\n", ty, st, us, m.objectId(), (double)m.type()); }
    }
    int64_t c[2]; gs_shell_counters(shell.shell(), c);
    std::printf("# frames run %lld gated %lld map %d loop_closed %d\n", (long long)c[0], (long long)c[1], gs_slam_map_size(gs_shell_slam(shell.shell())), gs_slam_loop_closed(gs_shell_slam(shell.shell())));
    return 0;
}
