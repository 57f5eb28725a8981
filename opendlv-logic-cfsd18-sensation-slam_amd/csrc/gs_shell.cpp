// gs_shell.cpp — the microservice shell around the Slam mirror, transport-independent.
//
// Mirrors reference src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119 and the timing glue of src/slam.cpp: the
// command line (the same --key=value keys, the same "at least 10 arguments" rule, :52), the seven data triggers with
// their senderStamp filters (:71-108), Slam::setUp's configuration (src/slam.cpp:736-756), the gathering window of a
// frame (initializeCollection, :221-257) and the keyframe gate (isKeyframe, :286-295), and what the localizer
// publishes (sendPose + sendCones, :404-410, 656-695).  Messages cross this boundary DECODED (type id, sender stamp,
// sample time, fields): the cluon / OD4 binding that decodes Envelopes and sends the outputs is csrc/gs_shell_cluon.cpp.
// The reference spawns a detached thread per frame that busy-waits gatheringTimeMs; here the owner of the shell calls
// gs_shell_poll with the wall clock, and a frame whose window has passed runs on the caller's thread.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "gs_internal.hpp"

using gs::fail;

enum { ID_WGS84 = 19, ID_ANGULAR_VELOCITY = 1031, ID_HEADING = 1051, ID_GEOLOCATION = 1116,
       ID_OBJECT_TYPE = 1131, ID_OBJECT_DIRECTION = 1133, ID_OBJECT_DISTANCE = 1134 };

struct gs_shell {
    gs_slam *slam = nullptr;
    std::map<std::string, std::string> args;
    uint32_t sender_stamp = 0, detect_cone_stamp = 0, estimation_stamp = 0;     // --id, --detectConeId, --estimationId
    uint32_t gathering_ms = 0; double time_between_keyframes = 0; int cones_per_packet = 0; int cid = 0;
    bool frame_open = false; int64_t frame_opened_us = 0, keyframe_us = 0;      // m_keyframeTimeStamp starts at zero: the first frame passes
    int64_t last_cone_sample_us = 0, yaw_sample_us = 0, geolocation_sample_us = 0;   // m_lastTimeStamp, m_yawReceivedTime, m_geolocationReceivedTime
    std::vector<gs_shell_msg> out;
    int64_t frames_run = 0, frames_gated = 0;
};

static const char *kUsage =
    " is a slam implementation for the CFSD18 project.\nUsage:   --cid=<OpenDaVINCI session> [--id=<Identifier in case of simulated units>] "
    "[--verbose] [Module specific parameters....]\nExample: --cid=111 --id=120 --detectConeId=118 --estimationId=114 --gatheringTimeMs=10 "
    "--sameConeThreshold=1.2 --refLatitude=48.123141 --refLongitude=12.34534 --timeBetweenKeyframes=0.5 --coneMappingThreshold=50 --conesPerPacket=20";

// cluon::getCommandlineArguments (reference src/cluon-complete-build.hpp:8398-8409): --key=value parameters, --flag -> "1",
// positional arguments -> "" (argv[0] included, as argh counts it)
static std::map<std::string, std::string> parse_args(int argc, const char *const *argv) {
    std::map<std::string, std::string> m;
    for (int i = 0; i < argc; ++i) { std::string a = argv[i] ? argv[i] : "";
        if (a.rfind("--", 0) == 0) { const size_t eq = a.find('=');
            if (eq == std::string::npos) m[a.substr(2)] = "1"; else m[a.substr(2, eq - 2)] = a.substr(eq + 1); }
        else m[a] = ""; }
    return m;
}
static bool num(const std::map<std::string, std::string> &m, const char *key, double &out) {
    auto it = m.find(key); if (it == m.end() || it->second.empty()) return false;
    char *end = nullptr; out = std::strtod(it->second.c_str(), &end); return end && *end == 0;
}

extern "C" int gs_shell_create(int32_t argc, const char *const *argv, int32_t device, gs_shell **out) {
    if (!out || argc < 0 || (argc > 0 && !argv)) return fail(GS_ERR_INVALID, "bad argument");
    *out = nullptr;
    auto args = parse_args(argc, argv);
    if (args.size() < 10) return fail(GS_ERR_INVALID, std::string(argc > 0 && argv[0] ? argv[0] : "opendlv-logic-cfsd18-sensation-slam") + kUsage);   // :52-56
    double cid, id = 0, detect, estim, gather, same, lat, lon, tbk, mapping, cpp;
    // std::stoi / std::stod of a missing key throw in the reference; here: an error code
    if (!num(args, "cid", cid) || !num(args, "detectConeId", detect) || !num(args, "estimationId", estim) || !num(args, "gatheringTimeMs", gather) ||
        !num(args, "sameConeThreshold", same) || !num(args, "refLatitude", lat) || !num(args, "refLongitude", lon) ||
        !num(args, "timeBetweenKeyframes", tbk) || !num(args, "coneMappingThreshold", mapping) || !num(args, "conesPerPacket", cpp) || !num(args, "id", id))
        return fail(GS_ERR_INVALID, "a required --key=value is missing or not a number (cid, id, detectConeId, estimationId, gatheringTimeMs, "
                                    "sameConeThreshold, refLatitude, refLongitude, timeBetweenKeyframes, coneMappingThreshold, conesPerPacket)");
    gs_config cfg; gs_config_default(&cfg);
    cfg.device = device; cfg.verbose = args.count("verbose") ? 1 : 0;
    cfg.same_cone_threshold = same; cfg.cone_mapping_threshold = mapping;       // Slam::setUp, src/slam.cpp:740,744
    cfg.reference_quirks = args.count("referenceQuirks") ? 1 : 0;               // not a reference key: SURVEY 8-B switches
    gs_slam *slam = nullptr;
    int rc = gs_slam_create(&cfg, &slam); if (rc != GS_OK) return rc;
    gs_shell *sh = new gs_shell();
    sh->slam = slam; sh->args = args; sh->cid = (int)cid;
    sh->sender_stamp = (uint32_t)id; sh->detect_cone_stamp = (uint32_t)detect; sh->estimation_stamp = (uint32_t)estim;
    sh->gathering_ms = (uint32_t)gather; sh->time_between_keyframes = tbk; sh->cones_per_packet = (int)cpp;
    gs_slam_set_gps_reference(slam, lat, lon);
    *out = sh;
    return GS_OK;
}
extern "C" int gs_shell_destroy(gs_shell *sh) { if (sh) { gs_slam_destroy(sh->slam); delete sh; } return GS_OK; }
extern "C" gs_slam *gs_shell_slam(gs_shell *sh) { return sh ? sh->slam : nullptr; }
extern "C" int gs_shell_cid(gs_shell *sh) { return sh ? sh->cid : fail(GS_ERR_INVALID, "null shell"); }

// The seven triggers (reference src/opendlv-logic-cfsd18-sensation-slam.cpp:102-108) behind their senderStamp filters (:71-100).
// Returns 1 if a trigger took the message, 0 if it was ignored (unknown type or foreign sender stamp), < 0 on error.
extern "C" int gs_shell_on_message(gs_shell *sh, const gs_shell_msg *m, int64_t now_us) {
    if (!sh || !m) return fail(GS_ERR_INVALID, "null argument");
    int rc = GS_OK, opened = 0;
    switch (m->data_type) {
        case ID_WGS84: if (m->sender_stamp != sh->estimation_stamp) return 0;
            rc = gs_slam_next_wgs84(sh->slam, m->v[0], m->v[1]); break;                    // nextSplitPose, position
        case ID_HEADING: if (m->sender_stamp != sh->estimation_stamp) return 0;
            rc = gs_slam_next_heading(sh->slam, m->v[0]); break;                             // nextSplitPose, heading
        case ID_GEOLOCATION: if (m->sender_stamp != sh->estimation_stamp) return 0;
            sh->geolocation_sample_us = m->sample_time_us;                                    // m_geolocationReceivedTime, src/slam.cpp:191
            rc = gs_slam_next_geolocation(sh->slam, m->v[0], m->v[1], m->v[2]); break;      // nextPose
        case ID_ANGULAR_VELOCITY: if (m->sender_stamp != sh->estimation_stamp) return 0;
            sh->yaw_sample_us = m->sample_time_us;                                            // m_yawReceivedTime, :216
            rc = gs_slam_next_yaw_rate(sh->slam, m->v[0]); break;                            // nextYawRate (angularVelocityZ)
        case ID_OBJECT_DIRECTION: if (m->sender_stamp != sh->detect_cone_stamp) return 0;
            sh->last_cone_sample_us = m->sample_time_us;                                      // m_lastTimeStamp, :73
            opened = rc = gs_slam_collect_direction(sh->slam, m->object_id, m->v[0], m->v[1]); break;
        case ID_OBJECT_DISTANCE: if (m->sender_stamp != sh->detect_cone_stamp) return 0;
            sh->last_cone_sample_us = m->sample_time_us;
            opened = rc = gs_slam_collect_distance(sh->slam, m->object_id, m->v[0]); break;
        case ID_OBJECT_TYPE: if (m->sender_stamp != sh->detect_cone_stamp) return 0;
            sh->last_cone_sample_us = m->sample_time_us;
            opened = rc = gs_slam_collect_type(sh->slam, m->object_id, (uint32_t)m->v[0]); break;
        default: return 0;
    }
    if (rc < 0) return rc;
    if (opened == 1) { sh->frame_open = true; sh->frame_opened_us = now_us; }   // the reference starts its gathering thread here (:94-95,120-121,145-146)
    return 1;
}

// The end of a gathering window: initializeCollection (:221-257) once gatheringTimeMs have passed since the frame's first
// message — extract + reset, the keyframe gate (:286-295, milliseconds against --timeBetweenKeyframes), performSLAM, and
// after loop closure what the localizer publishes.  Returns 1 if performSLAM ran, 0 if nothing was due or the frame was
// not a keyframe, < 0 on error.
extern "C" int gs_shell_poll(gs_shell *sh, int64_t now_us) {
    if (!sh) return fail(GS_ERR_INVALID, "null shell");
    if (!sh->frame_open || now_us - sh->frame_opened_us <= (int64_t)sh->gathering_ms * 1000) return 0;       // elapsed.count() > m_timeDiffMilliseconds*1000, :231
    sh->frame_open = false;
    int32_t k = 0; std::vector<double> cones(4 * 1000);
    int rc = gs_slam_collect_extract(sh->slam, &k, cones.data()); if (rc != GS_OK) return rc;
    if (k <= 0) return 0;
    const double elapsed_ms = std::fabs((double)(now_us - sh->keyframe_us)) / 1000.0;
    if (!(elapsed_ms > sh->time_between_keyframes)) { ++sh->frames_gated; return 0; }
    sh->keyframe_us = now_us;
    double odo[4];
    if ((rc = gs_slam_get_odometry(sh->slam, odo)) != GS_OK) return rc;
    if ((rc = gs_slam_set_sample_times(sh->slam, sh->yaw_sample_us, sh->last_cone_sample_us)) != GS_OK) return rc;
    const bool rejected = std::fabs(odo[0]) > 200 || std::fabs(odo[1]) > 200;          // performSLAM returns before anything else, :300-303
    if ((rc = gs_slam_perform(sh->slam, odo, cones.data(), k)) != GS_OK) return rc;
    ++sh->frames_run;
    // localizer: sendPose(); sendCones();  (:409-410) — every frame once the loop is closed (the closing one included), k > 1
    if (!rejected && gs_slam_loop_closed(sh->slam) == 1 && k > 1) {
        float p[3];
        if ((rc = gs_slam_encode_pose(sh->slam, p)) != GS_OK) return rc;
        gs_shell_msg o; std::memset(&o, 0, sizeof(o));
        o.sender_stamp = sh->sender_stamp; o.sample_time_us = sh->geolocation_sample_us;     // sampleTime = m_geolocationReceivedTime, :693,665
        o.data_type = ID_GEOLOCATION; o.v[0] = p[1]; o.v[1] = p[0]; o.v[2] = p[2];          // {latitude, longitude, heading} fields
        sh->out.push_back(o);
        const int n = sh->cones_per_packet;
        std::vector<float> az(n), di(n); std::vector<int32_t> ty(n);
        if (n > 0) { if ((rc = gs_slam_encode_cones(sh->slam, n, az.data(), di.data(), ty.data())) != GS_OK) return rc; }
        for (int i = 0; i < n; ++i) {
            o.object_id = (uint32_t)i;
            o.data_type = ID_OBJECT_DIRECTION; o.v[0] = az[i]; o.v[1] = 0; o.v[2] = 0; sh->out.push_back(o);
            o.data_type = ID_OBJECT_DISTANCE; o.v[0] = di[i]; sh->out.push_back(o);
            o.data_type = ID_OBJECT_TYPE; o.v[0] = ty[i]; sh->out.push_back(o);
        }
    }
    return 1;
}
extern "C" int gs_shell_take_output(gs_shell *sh, int32_t capacity, gs_shell_msg *out) {
    if (!sh || capacity < 0 || (capacity > 0 && !out)) return fail(GS_ERR_INVALID, "bad argument");
    const int n = (int)std::min<size_t>(sh->out.size(), (size_t)capacity);
    if (n > 0) std::memcpy(out, sh->out.data(), (size_t)n * sizeof(gs_shell_msg));
    sh->out.erase(sh->out.begin(), sh->out.begin() + n);
    return n;
}
extern "C" int gs_shell_pending_output(gs_shell *sh) { return sh ? (int)sh->out.size() : fail(GS_ERR_INVALID, "null shell"); }
extern "C" int gs_shell_counters(gs_shell *sh, int64_t out_run_gated[2]) {
    if (!sh || !out_run_gated) return fail(GS_ERR_INVALID, "null argument");
    out_run_gated[0] = sh->frames_run; out_run_gated[1] = sh->frames_gated; return GS_OK;
}
