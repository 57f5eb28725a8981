// gs_slam.cpp — host mirror of the graph side of class Slam (reference src/slam.hpp:43-137):
// performSLAM (src/slam.cpp:298-338), addPoseToGraph / addOdometryMeasurement (:433-459),
// addConesToMap (:552-635), addConeToGraph / addConeMeasurement (:525-550), loopClosing (:697-706),
// optimizeGraph (:461-484), updateMap (:713-732), localizer (:340-414), updatePoseFromGraph (:416-422).
//
// The per-frame control flow stays on the host exactly as in the reference; the arithmetic it calls
// (polar->XY, cone->global, the O(K*M) map scan, the optimiser) runs through the HIP kernels behind
// the C-ABI.  Message decode/encode (nextCone, sendCones, ...) stays above this boundary.
#include <array>
#include <cmath>
#include <cstring>
#include <vector>

#include "gs_internal.hpp"

struct MapCone { double x, y; int type, id; };              // reference src/cone.hpp:51-54

struct gs_slam {
    gs_config cfg{};
    gs_graph *g = nullptr;
    std::vector<MapCone> map;                                // m_map
    std::vector<std::array<double, 3>> poses;                // m_poses (raw odometry, never overwritten)
    int pose_id = 1000;                                      // m_poseId, reference src/slam.hpp:118
    bool loop_closing = false, loop_closing_complete = false;
    uint32_t current_cone_index = 0;                         // m_currentConeIndex
    double send_pose[3] = {0, 0, 0};                         // m_sendPose
    int optimise_calls = 0;
    // frame collector (m_coneCollector, m_lastObjectId, m_newFrame; reference src/slam.cpp:46, 67-152, 221-257)
    std::vector<double> collector = std::vector<double>(4 * 1000, 0.0);   // 4 x 1000, column-major: (az, zen, dist, type) per objectId
    uint32_t last_object_id = 0;
    bool new_frame = true;
    // odometry intake (m_gpsReference, m_odometryData, m_yawRate; reference src/slam.cpp:154-219)
    double gps_reference[2] = {0, 0};                        // latitude, longitude (degrees)
    double odometry[3] = {0, 0, 0};                          // x, y (metres from the reference), heading
    float yaw_rate = 0;                                      // m_yawRate is a float in the reference (src/slam.hpp:128)
    int64_t yaw_received_us = 0, last_cone_us = 0;           // m_yawReceivedTime, m_lastTimeStamp (sample times, microseconds)
};

using gs::fail;

extern "C" int gs_slam_create(const gs_config *cfg, gs_slam **out) {
    if (!out) return fail(GS_ERR_INVALID, "null out");
    *out = nullptr;
    gs_graph *g = nullptr;
    int rc = gs_create(cfg, &g);
    if (rc != GS_OK) return rc;
    if (!g->host_only) gs_reserve_device(g, (int64_t)24 << 20);      // the first optimizeGraph() of a lap-sized graph finds its device memory in place
    gs_slam *s = new gs_slam();
    s->g = g; s->cfg = g->cfg;
    *out = s;
    return GS_OK;
}
extern "C" int gs_slam_destroy(gs_slam *s) {
    if (!s) return GS_OK;
    gs_destroy(s->g);
    delete s;
    return GS_OK;
}
extern "C" gs_graph *gs_slam_graph(gs_slam *s) { return s ? s->g : nullptr; }
extern "C" int gs_slam_map_size(gs_slam *s) { return s ? (int)s->map.size() : fail(GS_ERR_INVALID, "null slam"); }
extern "C" int gs_slam_loop_closed(gs_slam *s) { return s ? (int)s->loop_closing_complete : fail(GS_ERR_INVALID, "null slam"); }
extern "C" int gs_slam_current_cone_index(gs_slam *s) { return s ? (int)s->current_cone_index : fail(GS_ERR_INVALID, "null slam"); }
extern "C" int gs_slam_get_send_pose(gs_slam *s, double out[3]) {
    if (!s || !out) return fail(GS_ERR_INVALID, "null argument");
    std::memcpy(out, s->send_pose, sizeof(s->send_pose));
    return GS_OK;
}
extern "C" int gs_slam_get_map(gs_slam *s, int32_t cap, double *xy, int32_t *type) {
    if (!s || !xy) return fail(GS_ERR_INVALID, "null argument");
    if (cap < (int)s->map.size()) return fail(GS_ERR_CAPACITY, "buffer too small");
    for (size_t j = 0; j < s->map.size(); ++j) { xy[2 * j] = s->map[j].x; xy[2 * j + 1] = s->map[j].y; if (type) type[j] = s->map[j].type; }
    return (int)s->map.size();
}

static double normalize_theta(double th) {
    if (th >= -M_PI && th < M_PI) return th;
    double m = std::floor(th / (2 * M_PI)); th -= m * 2 * M_PI;
    if (th >= M_PI) th -= 2 * M_PI;
    if (th < -M_PI) th += 2 * M_PI;
    return th;
}

// addConeMeasurement, reference src/slam.cpp:537-550 (z already converted by the A0 kernel)
static int add_cone_measurement(gs_slam *s, int cone_id, const double z_xy[2]) {
    const double w = s->cfg.cone_information;
    const double info[4] = {w, 0, 0, w};
    return gs_add_observation_edge(s->g, s->pose_id - 1, cone_id, z_xy, info);
}
// addConeToGraph, reference src/slam.cpp:525-535
static int add_cone_to_graph(gs_slam *s, const MapCone &c, const double z_xy[2]) {
    const double xy[2] = {c.x, c.y};
    int rc = gs_add_landmark(s->g, c.id, xy);
    if (rc != GS_OK) return rc;
    return add_cone_measurement(s, c.id, z_xy);
}
// optimizeGraph + updateMap, reference src/slam.cpp:461-484, 713-732
static int optimize_and_update_map(gs_slam *s) {
    int rc;
    if ((rc = gs_set_fixed_pose(s->g, 1000, 1)) != GS_OK) return rc;
    if ((rc = gs_set_fixed_pose(s->g, 1001, 1)) != GS_OK) return rc;
    if ((rc = gs_set_fixed_landmark(s->g, 0, 1)) != GS_OK) return rc;
    if ((rc = gs_set_fixed_landmark(s->g, 1, 1)) != GS_OK) return rc;
    rc = gs_optimize(s->g, s->cfg.optimize_iterations, nullptr);       // return value ignored by the reference (:481)
    ++s->optimise_calls;
    if (rc < 0) return rc;
    for (auto &c : s->map) { double xy[2]; if ((rc = gs_get_landmark(s->g, c.id, xy)) != GS_OK) return rc; c.x = xy[0]; c.y = xy[1]; }
    // the association map in HBM follows m_map: it holds every cone up to the ones added by earlier frames
    const int nres = gs_map_size(s->g);
    std::vector<double> xy(2 * (size_t)nres);
    for (int j = 0; j < nres; ++j) { xy[2 * j] = s->map[j].x; xy[2 * j + 1] = s->map[j].y; }
    return gs_map_set_xy(s->g, 0, nres, xy.data());
}

static double cone_distance(const MapCone &c, double x, double y) {     // distanceBetweenCones, :708-711
    return std::sqrt((c.x - x) * (c.x - x) + (c.y - y) * (c.y - y));
}

// appends the cones m_map gained since the last call to the association map in HBM
static int sync_resident_map(gs_slam *s) {
    const int have = gs_map_size(s->g), want = (int)s->map.size();
    if (have < 0) return have;
    if (want <= have) return GS_OK;
    std::vector<double> xy(2 * (size_t)(want - have)); std::vector<int32_t> ty(want - have);
    for (int j = have; j < want; ++j) { xy[2 * (size_t)(j - have)] = s->map[j].x; xy[2 * (size_t)(j - have) + 1] = s->map[j].y; ty[j - have] = s->map[j].type; }
    return gs_map_append(s->g, want - have, xy.data(), ty.data());
}

// One keyframe's front end: CoG-frame XY of every observation (the edge measurement), its global XY (the association
// query) and its first match in the map as it stood at the start of the frame — ONE fused launch against the map that
// stays resident in HBM (gs_frame_frontend), one wait.  Under reference_quirks one more "observation" rides along: the
// pose itself, which the reference's localizer passes where (azimuth, zenith, distance) is expected.
struct FrameXY { std::vector<double> zxy, gxy; std::vector<int32_t> idx; double quirk_z[2] = {0, 0}; };
static int frame_frontend(gs_slam *s, const double pose[3], const double *cones, int k, bool localizer_test, FrameXY &fx) {
    const bool quirks = s->cfg.reference_quirks != 0;
    const int kk = k + (quirks ? 1 : 0);
    std::vector<double> obs(4 * (size_t)kk);
    std::memcpy(obs.data(), cones, 4 * (size_t)k * sizeof(double));
    if (quirks) { obs[4 * (size_t)k] = pose[0]; obs[4 * (size_t)k + 1] = pose[1]; obs[4 * (size_t)k + 2] = pose[2]; obs[4 * (size_t)k + 3] = 0; }
    fx.zxy.resize(2 * (size_t)kk); fx.gxy.resize(2 * (size_t)kk); fx.idx.resize(kk);
    // the localizer's type test has no fabs in the reference (src/slam.cpp:360): kept only under reference_quirks
    int rc = gs_frame_frontend(s->g, pose, obs.data(), kk, s->cfg.same_cone_threshold, 1e-4, localizer_test && quirks ? 1 : 0,
                               fx.zxy.data(), fx.gxy.data(), fx.idx.data());
    if (rc != GS_OK) return rc;
    if (quirks) { fx.quirk_z[0] = fx.zxy[2 * (size_t)k]; fx.quirk_z[1] = fx.zxy[2 * (size_t)k + 1]; }
    return GS_OK;
}

// addConesToMap, reference src/slam.cpp:552-635
static int add_cones_to_map(gs_slam *s, const double pose[3], const double *cones, int k, const FrameXY &fx) {
    int rc;
    const std::vector<double> &zxy = fx.zxy, &gxy = fx.gxy;
    const bool quirks = s->cfg.reference_quirks != 0;
    const int m0 = gs_map_size(s->g);                                // the map the association kernel saw (= m_map at the start of the frame)
    int first = 0;
    if (s->map.empty()) {                                            // :554-567
        MapCone c{gxy[0], gxy[1], (int)cones[3], 0};
        s->map.push_back(c);
        if ((rc = add_cone_to_graph(s, c, &zxy[0])) != GS_OK) return rc;
        if (!quirks) first = 1;                     // the reference re-matches i = 0 and adds the edge twice
    }
    double min_distance = 100;
    bool optimise_pending = false;
    for (int i = first; i < k; ++i) {                                // :570-634
        const double d2car = cones[4 * i + 2], type_i = cones[4 * i + 3];
        bool found = false; int j = -1;
        if (!s->loop_closing) {                                      // the while loop's guard, :575
            if (fx.idx[i] >= 0) { found = true; j = fx.idx[i]; }     // A1 on the device: first match among the cones of earlier frames
            else for (int t = m0; t < (int)s->map.size(); ++t)       // cones appended earlier in this frame
                if (std::fabs(s->map[t].type - type_i) < 1e-4 && cone_distance(s->map[t], gxy[2 * i], gxy[2 * i + 1]) < s->cfg.same_cone_threshold) { found = true; j = t; break; }
        }
        if (found) {
            if ((rc = add_cone_measurement(s, s->map[j].id, &zxy[2 * i])) != GS_OK) return rc;      // :591
            // loopClosing(), :697-706, asked with m_currentConeIndex as it was BEFORE this match (:593 precedes :598)
            if (cone_distance(s->map[0], s->map[j].x, s->map[j].y) < s->cfg.loop_closing_radius &&
                s->current_cone_index > (uint32_t)s->cfg.loop_closing_min_index && d2car < s->cfg.cone_mapping_threshold && !s->loop_closing)
                s->loop_closing = true;
            if (d2car < min_distance) { s->current_cone_index = (uint32_t)j; min_distance = d2car; }
        }
        if (d2car < s->cfg.cone_mapping_threshold && !found && !s->loop_closing) {                 // :608-623
            MapCone c{gxy[2 * i], gxy[2 * i + 1], (int)type_i, (int)s->map.size()};
            s->map.push_back(c);
            if ((rc = add_cone_to_graph(s, c, &zxy[2 * i])) != GS_OK) return rc;
        }
        if (s->loop_closing) {                                       // :625-633
            if (quirks) { if ((rc = sync_resident_map(s)) != GS_OK) return rc;
                          if ((rc = optimize_and_update_map(s)) != GS_OK) return rc; s->loop_closing_complete = true; }   // §8-B.2: once per remaining observation
            else optimise_pending = true;
        }
    }
    if ((rc = sync_resident_map(s)) != GS_OK) return rc;             // this frame's new cones join the map in HBM (one asynchronous copy)
    if (optimise_pending) { if ((rc = optimize_and_update_map(s)) != GS_OK) return rc; s->loop_closing_complete = true; }
    return GS_OK;
}

// localizer, reference src/slam.cpp:340-414: re-associate against the frozen map, one more edge per re-observed cone,
// the send pose is the raw estimate of the newest pose vertex (optimizeGraph is commented out there, :403)
static int localizer(gs_slam *s, const double *cones, int k, const FrameXY &fx) {
    int rc;
    const bool quirks = s->cfg.reference_quirks != 0;
    uint32_t current = s->current_cone_index; double min_distance = 100; int reobserved = 0;
    for (int i = 0; i < k; ++i) {
        const double d2car = cones[4 * i + 2];
        const int j = fx.idx[i];                                     // A1 on the device: the while loop of :355-382 against the whole map
        if (j < 0) continue;
        ++reobserved;
        // §8-B.4: the reference passes the POSE where (az, zen, dist) is expected (:373)
        const double *z = quirks ? fx.quirk_z : &fx.zxy[2 * (size_t)i];
        if ((rc = add_cone_measurement(s, s->map[j].id, z)) != GS_OK) return rc;
        if (d2car < min_distance) { current = (uint32_t)j; min_distance = d2car; }
    }
    if (reobserved > 0) s->current_cone_index = current;            // :387 (uninitialised in the reference when nothing matched)
    // updatePoseFromGraph (:416-422)
    return gs_get_pose(s->g, s->pose_id - 1, s->send_pose);
}

extern "C" int gs_slam_perform(gs_slam *s, const double odometry[3], const double *cones, int32_t k) {
    if (!s || !odometry || k < 0 || (k > 0 && !cones)) return fail(GS_ERR_INVALID, "bad argument");
    if (std::fabs(odometry[0]) > 200 || std::fabs(odometry[1]) > 200) return GS_OK;      // :300-303
    // heading compensated by the yaw rate over the time between the last yaw-rate and the last cone message (:306-318)
    double pose[3] = {odometry[0], odometry[1], odometry[2]};
    const double elapsed = std::fabs((double)(s->yaw_received_us - s->last_cone_us)) / 1000000.0;
    if (elapsed > 0 && elapsed < 1) pose[2] = pose[2] - (double)s->yaw_rate * elapsed;
    s->poses.push_back({pose[0], pose[1], pose[2]});
    int rc;
    const auto optimised_before = s->optimise_calls;
    // cfg.optimize_every_keyframe (not the reference's behaviour: its calls at :403, :594, :620-621 are commented out): optimizeGraph +
    // updateMap at the end of a keyframe that did not run them itself, once the gauge vertices and something to solve for exist
    auto keyframe_optimise = [&]() -> int {
        if (!s->cfg.optimize_every_keyframe || s->optimise_calls != optimised_before || s->pose_id - 1000 < 3 || s->map.size() < 3) return GS_OK;
        int r = optimize_and_update_map(s); if (r < 0) return r;
        if (s->loop_closing_complete) return gs_get_pose(s->g, s->pose_id - 1, s->send_pose);      // updatePoseFromGraph again: the optimised pose goes out
        return GS_OK; };
    // ---- addPoseToGraph + addOdometryMeasurement (:433-459)
    if ((rc = gs_add_pose(s->g, s->pose_id, pose)) != GS_OK) return rc;
    if (s->pose_id > 1000) {
        double prev[3];
        if ((rc = gs_get_pose(s->g, s->pose_id - 1, prev)) != GS_OK) return rc;
        double th = normalize_theta(-prev[2]), c = std::cos(th), sn = std::sin(th);
        double ix = c * (-prev[0]) - sn * (-prev[1]), iy = sn * (-prev[0]) + c * (-prev[1]);
        double z[3] = {ix + (c * pose[0] - sn * pose[1]), iy + (sn * pose[0] + c * pose[1]), normalize_theta(th + pose[2])};
        const double w = s->cfg.odometry_information;
        const double info[9] = {w, 0, 0, 0, w, 0, 0, 0, w};
        if ((rc = gs_add_odometry_edge(s->g, s->pose_id - 1, s->pose_id, z, info)) != GS_OK) return rc;
    }
    s->pose_id++;
    if (k == 0) return GS_OK;          // initializeCollection never passes an empty frame (:245)

    // two independent ifs in the reference (:329-334): the frame that completes the loop closure ALSO runs the localizer,
    // against the map updateMap has just rewritten (a second front-end launch, in that one frame only)
    FrameXY fx;
    if (!s->loop_closing_complete) {
        if ((rc = frame_frontend(s, pose, cones, k, false, fx)) != GS_OK) return rc;
        if ((rc = add_cones_to_map(s, pose, cones, k, fx)) != GS_OK) return rc;
        if (s->loop_closing_complete && k > 1) {
            if ((rc = frame_frontend(s, pose, cones, k, true, fx)) != GS_OK) return rc;
            if ((rc = localizer(s, cones, k, fx)) != GS_OK) return rc; }
    } else if (k > 1) {
        if ((rc = frame_frontend(s, pose, cones, k, true, fx)) != GS_OK) return rc;
        if ((rc = localizer(s, cones, k, fx)) != GS_OK) return rc;
    }
    return keyframe_optimise();
}

// ------------------------------------------------------------------ f-2: frame collector and output encoders
// Slam::nextCone (reference src/slam.cpp:67-152): one message = one field of column objectId of the 4 x 1000 collector;
// m_lastObjectId = max; the first message after a flush opens the frame (m_newFrame true -> false), which in the
// reference starts the waiting thread that ends in initializeCollection.  Return value: 1 if this message opened a
// frame, 0 otherwise, < 0 on error.  objectId >= 1000 is an error here (the reference indexes unchecked).
static int collect(gs_slam *s, uint32_t id, int row0, const double *vals, int nvals) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    if (id >= 1000) return fail(GS_ERR_INVALID, "objectId beyond the 4 x 1000 collector");
    for (int r = 0; r < nvals; ++r) s->collector[4 * (size_t)id + row0 + r] = vals[r];
    s->last_object_id = std::max(s->last_object_id, id);
    const int opened = s->new_frame ? 1 : 0;
    s->new_frame = false;
    return opened;
}
extern "C" int gs_slam_collect_direction(gs_slam *s, uint32_t object_id, double azimuth_deg, double zenith_deg) {
    const double v[2] = {azimuth_deg, zenith_deg}; return collect(s, object_id, 0, v, 2);
}
extern "C" int gs_slam_collect_distance(gs_slam *s, uint32_t object_id, double distance) { return collect(s, object_id, 2, &distance, 1); }
extern "C" int gs_slam_collect_type(gs_slam *s, uint32_t object_id, uint32_t type) { const double v = (double)type; return collect(s, object_id, 3, &v, 1); }
// Slam::initializeCollection (reference src/slam.cpp:221-257) after its wait and without the keyframe gate (both are
// transport timing, out of scope): take the leftmost m_lastObjectId + 1 columns, reset the collector, run performSLAM.
extern "C" int gs_slam_collect_extract(gs_slam *s, int32_t *k_out, double *cones_out_4xk) {
    if (!s || !k_out || !cones_out_4xk) return fail(GS_ERR_INVALID, "null argument");
    const int k = (int)s->last_object_id + 1;
    std::memcpy(cones_out_4xk, s->collector.data(), 4 * (size_t)k * sizeof(double));       // extractedCones = leftCols(m_lastObjectId + 1), :241
    s->new_frame = true; s->last_object_id = 0;
    std::fill(s->collector.begin(), s->collector.end(), 0.0);
    *k_out = k;
    return GS_OK;
}
extern "C" int gs_slam_collect_flush(gs_slam *s, const double pose_xytheta[3], int32_t *k_out, double *cones_out_4xk) {
    if (!s || !pose_xytheta) return fail(GS_ERR_INVALID, "null argument");
    int32_t k = 0; std::vector<double> extracted(4 * 1000);
    int rc = gs_slam_collect_extract(s, &k, extracted.data()); if (rc != GS_OK) return rc;
    if (k_out) *k_out = k;
    if (cones_out_4xk) std::memcpy(cones_out_4xk, extracted.data(), 4 * (size_t)k * sizeof(double));
    return gs_slam_perform(s, pose_xytheta, extracted.data(), k);      // extractedCones.cols() > 0 always holds (:245)
}
// Slam::sendCones + Cone::getDirection / getDistance (reference src/slam.cpp:656-677, src/cone.cpp:34-53): the
// conesPerPacket map cones starting at m_currentConeIndex, wrapping around the map, seen from m_sendPose; message
// fields are float32.  Quirk 8-B.7 (the radian heading scaled by 1/RAD2DEG before it is subtracted from degrees) is
// reproduced under cfg.reference_quirks, otherwise the heading is converted to degrees.
// One cone as the three messages carry it: Cone::getDirection / Cone::getDistance (reference src/cone.cpp:34-53), float32
// fields.  reference_quirks != 0: the reference's arithmetic as written — the radian heading scaled by 1 / RAD2DEG before
// it is subtracted from degrees; 0: the heading converted to degrees.  Stateless (no handle, no device).
extern "C" int gs_cone_encode(const double cone_xy[2], const double pose_xytheta[3], int32_t reference_quirks, float *azimuth_deg, float *distance) {
    if (!cone_xy || !pose_xytheta || !azimuth_deg || !distance) return fail(GS_ERR_INVALID, "null argument");
    const double RAD2DEG = 57.295779513082325;                       // reference src/cone.hpp:55, src/slam.hpp:135
    const double x = cone_xy[0] - pose_xytheta[0], y = cone_xy[1] - pose_xytheta[1];
    const double heading = reference_quirks ? pose_xytheta[2] * (1 / RAD2DEG) : pose_xytheta[2] * RAD2DEG;
    *azimuth_deg = (float)(std::atan2(y, x) * RAD2DEG - heading);
    *distance = (float)std::sqrt(x * x + y * y);
    return GS_OK;
}
extern "C" int gs_slam_encode_cones(gs_slam *s, int32_t cones_per_packet, float *azimuth_deg, float *distance, int32_t *type) {
    if (!s || cones_per_packet < 0 || (cones_per_packet > 0 && (!azimuth_deg || !distance || !type))) return fail(GS_ERR_INVALID, "bad argument");
    if (cones_per_packet > 0 && s->map.empty()) return fail(GS_ERR_INVALID, "empty map");
    const size_t n = s->map.size();
    for (int i = 0; i < cones_per_packet; ++i) {
        size_t index = s->current_cone_index + (size_t)i;
        if (index >= n) index -= n;                                  // the reference's single wrap (:666-667) ...
        if (index >= n) index %= n;                                  // ... made safe for conesPerPacket > map size (reference: out of bounds)
        const MapCone &c = s->map[index];
        const double xy[2] = {c.x, c.y};
        gs_cone_encode(xy, s->send_pose, s->cfg.reference_quirks, &azimuth_deg[i], &distance[i]);
        type[i] = c.type;
    }
    return GS_OK;
}

// ------------------------------------------------------------------ f-4: odometry intake and pose output
// m_gpsReference comes from the command line in the reference (src/opendlv-logic-cfsd18-sensation-slam.cpp, keys
// refLatitude / refLongitude); latitude first, as wgs84::toCartesian takes it.
extern "C" int gs_slam_set_gps_reference(gs_slam *s, double latitude_deg, double longitude_deg) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    s->gps_reference[0] = latitude_deg; s->gps_reference[1] = longitude_deg;
    return GS_OK;
}
// Slam::nextSplitPose, GeodeticWgs84Reading branch (reference src/slam.cpp:156-176): position only
extern "C" int gs_slam_next_wgs84(gs_slam *s, double latitude_deg, double longitude_deg) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    const double p[2] = {latitude_deg, longitude_deg};
    return gs_wgs84_to_cartesian(s->gps_reference, p, s->odometry);
}
// Slam::nextSplitPose, GeodeticHeadingReading branch (:177-184): north heading -> [-PI, PI] with the reference's
// float-literal PI (src/slam.hpp:136)
extern "C" int gs_slam_next_heading(gs_slam *s, double north_heading) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    const double PI = 3.14159265f;
    double h = north_heading - PI;
    h = (h > PI) ? (h - 2 * PI) : h;
    h = (h < -PI) ? (h + 2 * PI) : h;
    s->odometry[2] = h;
    return GS_OK;
}
// Slam::nextPose (:187-209): a Geolocation message carries position and heading together (heading taken as is)
extern "C" int gs_slam_next_geolocation(gs_slam *s, double latitude_deg, double longitude_deg, double heading) {
    int rc = gs_slam_next_wgs84(s, latitude_deg, longitude_deg);
    if (rc != GS_OK) return rc;
    s->odometry[2] = heading;
    return GS_OK;
}
// Slam::nextYawRate (:211-219)
extern "C" int gs_slam_next_yaw_rate(gs_slam *s, double angular_velocity_z) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    s->yaw_rate = (float)angular_velocity_z / 4;              // float arithmetic as in the reference (:214)
    return GS_OK;
}
// m_yawReceivedTime (:216) and m_lastTimeStamp (:73,102,129): sample times of the last yaw-rate message and of the last
// cone message, microseconds.  performSLAM turns their distance into the heading compensation (:306-318).
extern "C" int gs_slam_set_sample_times(gs_slam *s, int64_t yaw_received_us, int64_t last_cone_us) {
    if (!s) return fail(GS_ERR_INVALID, "null handle");
    s->yaw_received_us = yaw_received_us; s->last_cone_us = last_cone_us;
    return GS_OK;
}
extern "C" int gs_slam_get_odometry(gs_slam *s, double out_xy_heading_yawrate[4]) {
    if (!s || !out_xy_heading_yawrate) return fail(GS_ERR_INVALID, "null argument");
    std::memcpy(out_xy_heading_yawrate, s->odometry, 3 * sizeof(double)); out_xy_heading_yawrate[3] = s->yaw_rate;
    return GS_OK;
}
// Slam::sendPose (:679-695): the send pose back to WGS84, float32 message fields {longitude, latitude, heading}.
// fromCartesian returns {latitude, longitude}; the reference writes element 0 into `longitude` and element 1 into
// `latitude` — kept under cfg.reference_quirks, put right otherwise.
extern "C" int gs_slam_encode_pose(gs_slam *s, float out_lon_lat_heading[3]) {
    if (!s || !out_lon_lat_heading) return fail(GS_ERR_INVALID, "null argument");
    double latlon[2];
    int rc = gs_wgs84_from_cartesian(s->gps_reference, s->send_pose, latlon);
    if (rc != GS_OK) return rc;
    const bool q = s->cfg.reference_quirks != 0;
    out_lon_lat_heading[0] = (float)(q ? latlon[0] : latlon[1]);
    out_lon_lat_heading[1] = (float)(q ? latlon[1] : latlon[0]);
    out_lon_lat_heading[2] = (float)s->send_pose[2];
    return GS_OK;
}
