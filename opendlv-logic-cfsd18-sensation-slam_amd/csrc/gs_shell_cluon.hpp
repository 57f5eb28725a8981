// gs_shell_cluon.hpp — binds the transport-independent shell (gs_shell.cpp, include/graphslam.h) to libcluon:
// cluon::data::Envelope in (decoded with cluon::extractMessage into the shell's message record), the opendlv messages
// the localizer publishes out (reference src/slam.cpp:656-695) through a sender callback (OD4Session::send in the
// microservice, a capture in the test driver).
//
// Needs the reference's single-header libcluon and the message set its generator produces from the .odvd — neither is
// copied into this repository: oracle/Makefile (target ref_shell) compiles against them where they lie
// in the reference tree, with the generated message set written to the git-ignored oracle/_ref/.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

#include "cluon-complete.hpp"
#include "opendlv-standard-message-set.hpp"

#include "../../include/graphslam.h"

class ShellCluon {
   public:
    // out: one decoded output message of the shell (type id as in gs_shell_msg) to be sent with this sample time / sender stamp
    using Sender = std::function<void(const gs_shell_msg &)>;

    ShellCluon(int argc, char **argv, int device, Sender sender) : m_sender(std::move(sender)) {
        m_rc = gs_shell_create(argc, argv, device, &m_shell);
    }
    ~ShellCluon() { gs_shell_destroy(m_shell); }
    ShellCluon(const ShellCluon &) = delete;
    ShellCluon &operator=(const ShellCluon &) = delete;
    int status() const { return m_rc; }
    gs_shell *shell() { return m_shell; }

    // one Envelope off the wire: the field decode each trigger of the reference does with cluon::extractMessage
    // (src/slam.cpp:74,103,130,157,178,192,214); the senderStamp filters live in gs_shell_on_message
    int onEnvelope(cluon::data::Envelope &&env, int64_t now_us) {
        gs_shell_msg m{};
        m.data_type = env.dataType(); m.sender_stamp = env.senderStamp();
        m.sample_time_us = cluon::time::toMicroseconds(env.sampleTimeStamp());
        const int32_t dt = env.dataType();                       // (the generated ID() functions are not constexpr: no switch)
        if (dt == opendlv::proxy::GeodeticWgs84Reading::ID()) { auto x = cluon::extractMessage<opendlv::proxy::GeodeticWgs84Reading>(std::move(env));
                m.v[0] = x.latitude(); m.v[1] = x.longitude(); }
        else if (dt == opendlv::proxy::GeodeticHeadingReading::ID()) { auto x = cluon::extractMessage<opendlv::proxy::GeodeticHeadingReading>(std::move(env));
                m.v[0] = x.northHeading(); }
        else if (dt == opendlv::logic::sensation::Geolocation::ID()) { auto x = cluon::extractMessage<opendlv::logic::sensation::Geolocation>(std::move(env));
                m.v[0] = x.latitude(); m.v[1] = x.longitude(); m.v[2] = x.heading(); }
        else if (dt == opendlv::proxy::AngularVelocityReading::ID()) { auto x = cluon::extractMessage<opendlv::proxy::AngularVelocityReading>(std::move(env));
                m.v[0] = x.angularVelocityZ(); }
        else if (dt == opendlv::logic::perception::ObjectDirection::ID()) { auto x = cluon::extractMessage<opendlv::logic::perception::ObjectDirection>(std::move(env));
                m.object_id = x.objectId(); m.v[0] = x.azimuthAngle(); m.v[1] = x.zenithAngle(); }
        else if (dt == opendlv::logic::perception::ObjectDistance::ID()) { auto x = cluon::extractMessage<opendlv::logic::perception::ObjectDistance>(std::move(env));
                m.object_id = x.objectId(); m.v[0] = x.distance(); }
        else if (dt == opendlv::logic::perception::ObjectType::ID()) { auto x = cluon::extractMessage<opendlv::logic::perception::ObjectType>(std::move(env));
                m.object_id = x.objectId(); m.v[0] = x.type(); }
        else return 0;
        return gs_shell_on_message(m_shell, &m, now_us);
    }

    // runs a frame whose gathering window has passed and hands what it publishes to the sender
    int poll(int64_t now_us) {
        const int rc = gs_shell_poll(m_shell, now_us);
        if (rc < 0) return rc;
        const int n = gs_shell_pending_output(m_shell);
        if (n > 0) { std::vector<gs_shell_msg> out((size_t)n);
            const int got = gs_shell_take_output(m_shell, n, out.data());
            for (int i = 0; i < got; ++i) m_sender(out[(size_t)i]); }
        return rc;
    }

    // the typed send the reference does with od4.send(message, sampleTime, senderStamp) (src/slam.cpp:668-676, 693)
    template <class Od4> static void sendWith(Od4 &od4, const gs_shell_msg &o) {
        const cluon::data::TimeStamp ts = cluon::time::fromMicroseconds(o.sample_time_us);
        if (o.data_type == opendlv::logic::sensation::Geolocation::ID()) { opendlv::logic::sensation::Geolocation m;
                m.latitude(o.v[0]); m.longitude(o.v[1]); m.heading(static_cast<float>(o.v[2])); od4.send(m, ts, o.sender_stamp); }
        else if (o.data_type == opendlv::logic::perception::ObjectDirection::ID()) { opendlv::logic::perception::ObjectDirection m;
                m.objectId(o.object_id); m.azimuthAngle(static_cast<float>(o.v[0])); m.zenithAngle(static_cast<float>(o.v[1])); od4.send(m, ts, o.sender_stamp); }
        else if (o.data_type == opendlv::logic::perception::ObjectDistance::ID()) { opendlv::logic::perception::ObjectDistance m;
                m.objectId(o.object_id); m.distance(static_cast<float>(o.v[0])); od4.send(m, ts, o.sender_stamp); }
        else if (o.data_type == opendlv::logic::perception::ObjectType::ID()) { opendlv::logic::perception::ObjectType m;
                m.objectId(o.object_id); m.type(static_cast<uint32_t>(o.v[0])); od4.send(m, ts, o.sender_stamp); }
    }

   private:
    gs_shell *m_shell = nullptr;
    int m_rc = 0;
    Sender m_sender;
};
