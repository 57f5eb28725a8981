"""Test-only restatement, in Python over the CPU oracle, of the graph side of the reference's class Slam:
frame collector nextCone / initializeCollection (reference src/slam.cpp:67-152, 221-257), output encoders sendCones +
Cone::getDirection / getDistance (src/slam.cpp:656-677, src/cone.cpp:34-53), performSLAM (reference src/slam.cpp:298-338), addPoseToGraph/addOdometryMeasurement (:433-459), addConesToMap
(:552-635), addConeToGraph/addConeMeasurement (:525-550), loopClosing (:697-706), optimizeGraph (:461-484),
updateMap (:713-732), localizer (:340-414), updatePoseFromGraph (:416-422).

It is the checker for the product's host mirror csrc/gs_slam.cpp (C++ over the HIP C-ABI).  perform / _add_cones_to_map /
_localizer follow the reference's text statement by statement (line numbers in the comments), with the oracle's A0 and
the oracle's optimiser underneath; tests/test_gpu_parity.py additionally asserts facts read off the reference directly
(edges added in the loop-closing frame, the pose published there) so that the product is not only compared with this twin.  `quirks=False` is the
documented clean mode (SURVEY §8-B items 1, 2, 4 off); `quirks=True` keeps them."""
import numpy as np

from oracle import pyoracle as po


class RefSlam:
    def __init__(self, same_cone_threshold=1.2, cone_mapping_threshold=67.0, quirks=False, iterations=10, optimize_every_keyframe=False):
        self.g = po.OracleGraph(); self.fe = po.OracleFrontend()
        self.thr = same_cone_threshold; self.map_thr = cone_mapping_threshold
        self.quirks = quirks; self.iterations = iterations
        self.optimize_every_keyframe = optimize_every_keyframe     # NOT the reference: its optimizeGraph() calls at :403, :594, :620-621 are commented out
        self.map = []                      # [x, y, type, id]
        self.n_poses = 0
        self.loop_closing = False; self.loop_closing_complete = False
        self.current_cone_index = 0
        self.send_pose = np.zeros(3)
        self.optimise_calls = 0
        self.collector = np.zeros((4, 1000)); self.last_object_id = 0; self.new_frame = True
        self.poses = []                    # m_poses
        self.yaw_rate = np.float32(0.0); self.yaw_received_us = 0; self.last_cone_us = 0
        self.localizer_calls = 0

    # --- frame collector (reference src/slam.cpp:67-152, 221-257; the wait and the keyframe gate are transport timing)
    def _collect(self, object_id, rows, vals):
        self.collector[rows, object_id] = vals
        self.last_object_id = max(self.last_object_id, object_id)
        opened = self.new_frame; self.new_frame = False
        return int(opened)

    def collect_direction(self, object_id, az, zen): return self._collect(object_id, [0, 1], [az, zen])
    def collect_distance(self, object_id, dist): return self._collect(object_id, [2], [dist])
    def collect_type(self, object_id, ty): return self._collect(object_id, [3], [float(ty)])

    def collect_flush(self, pose):
        extracted = self.collector[:, :self.last_object_id + 1].T.copy()
        self.new_frame = True; self.last_object_id = 0; self.collector[:] = 0.0
        self.perform(pose, extracted)
        return extracted

    # --- output encoders (reference src/slam.cpp:656-677, src/cone.cpp:34-53); float32 message fields
    def encode_cones(self, cones_per_packet):
        RAD2DEG = 57.295779513082325
        n = len(self.map); az = np.zeros(cones_per_packet, np.float32); di = np.zeros(cones_per_packet, np.float32); ty = np.zeros(cones_per_packet, np.int32)
        for i in range(cones_per_packet):
            idx = self.current_cone_index + i
            if idx >= n: idx -= n
            if idx >= n: idx %= n
            c = self.map[idx]
            x, y = c[0] - self.send_pose[0], c[1] - self.send_pose[1]
            heading = self.send_pose[2] * (1 / RAD2DEG) if self.quirks else self.send_pose[2] * RAD2DEG
            az[i] = np.float32(np.arctan2(y, x) * RAD2DEG - heading); di[i] = np.float32(np.sqrt(x * x + y * y)); ty[i] = c[2]
        return az, di, ty

    # --- helpers on the oracle graph (indices: pose k = id 1000 + k)
    def _add_measurement(self, cone_id, z):
        self.g.add_observation_edges([self.n_poses - 1], [cone_id], [z], (0.01 * np.eye(2)).reshape(1, 4))

    def _add_cone(self, xy, ctype, z):
        cid = len(self.map)
        self.map.append([xy[0], xy[1], int(ctype), cid])
        self.g.add_landmarks([xy])
        self._add_measurement(cid, z)

    def _optimise(self):
        self.g.set_fixed_pose(0); self.g.set_fixed_pose(1); self.g.set_fixed_landmark(0); self.g.set_fixed_landmark(1)
        self.g.optimize(self.iterations, ordering=1)
        self.optimise_calls += 1
        L = self.g.landmarks()
        for c in self.map:
            c[0], c[1] = L[c[3]]

    # --- odometry side inputs of performSLAM (reference src/slam.cpp:211-219, :73,102,129)
    def next_yaw_rate(self, wz):
        self.yaw_rate = np.float32(np.float32(wz) / np.float32(4))        # float m_yawRate (src/slam.hpp:128)

    def set_sample_times(self, yaw_received_us, last_cone_us):
        self.yaw_received_us, self.last_cone_us = int(yaw_received_us), int(last_cone_us)

    # --- performSLAM, reference src/slam.cpp:298-338, statement by statement
    def perform(self, odometry, cones):
        odometry = np.asarray(odometry, dtype=np.float64); cones = np.asarray(cones, dtype=np.float64).reshape(-1, 4)
        if abs(odometry[0]) > 200 or abs(odometry[1]) > 200:              # :300-303
            return
        pose = odometry.copy()                                            # :307
        elapsed = abs(float(self.yaw_received_us - self.last_cone_us)) / 1000000.0     # :309
        if 0 < elapsed < 1:                                               # :315-317
            pose[2] = pose[2] - float(self.yaw_rate) * elapsed
        self.poses.append(pose.copy())                                    # :319
        self._add_pose_to_graph(pose)                                     # :325
        if len(cones) == 0:                                               # never happens in the reference (:245)
            return
        before = self.optimise_calls
        if not self.loop_closing_complete:                                # :329-331
            self._add_cones_to_map(cones, pose)
        if self.loop_closing_complete and len(cones) > 1:                 # :332-334 — a second, independent if
            self._localizer(pose, cones)
        if self.optimize_every_keyframe and self.optimise_calls == before and self.n_poses >= 3 and len(self.map) >= 3:
            self._optimise()                                              # the commented-out calls, once per keyframe
            if self.loop_closing_complete:
                self.send_pose = self.g.poses()[self.n_poses - 1].copy()

    # addPoseToGraph + addOdometryMeasurement, :433-459
    def _add_pose_to_graph(self, pose):
        self.g.add_poses([pose])
        if self.n_poses > 0:                                              # m_poseId > 1000
            prev = self.g.poses()[self.n_poses - 1]
            L = po.lib(); inv = np.zeros(3); z = np.zeros(3)
            L.orc_se2_inverse(po._d(np.ascontiguousarray(prev)), po._d(inv)); L.orc_se2_compose(po._d(inv), po._d(pose.copy()), po._d(z))
            self.g.add_odometry_edges([self.n_poses - 1], [self.n_poses], [z], (5 * np.eye(3)).reshape(1, 9))
        self.n_poses += 1

    # addConesToMap, :552-635
    def _add_cones_to_map(self, cones, pose):
        K = len(cones)
        zxy = self.fe.polar_to_xy(cones[:, 0], cones[:, 1], cones[:, 2])          # Spherical2Cartesian of (az, zen, dist), :539
        gxy = self.fe.cone_to_global(pose[None], np.zeros(K, dtype=np.int32), cones)   # coneToGlobal, :572
        first = 0
        if not self.map:                                                  # :554-567
            self._add_cone(gxy[0], int(cones[0, 3]), zxy[0])
            if not self.quirks:
                first = 1                                                 # clean mode drops the duplicate edge (§8-B.1)
        min_distance = 100.0; pending = False
        for i in range(first, K):                                         # :570
            d2car, ty = cones[i, 2], cones[i, 3]
            found, j = False, 0
            while not found and j < len(self.map) and not self.loop_closing:      # :575
                c = self.map[j]
                if abs(c[2] - ty) < 0.0001:                               # :576
                    if np.sqrt((c[0] - gxy[i, 0]) ** 2 + (c[1] - gxy[i, 1]) ** 2) < self.thr:       # :579, :584
                        found = True
                        self._add_measurement(c[3], zxy[i])               # :591
                        if self._loop_closing_candidate(c, d2car) and not self.loop_closing:       # :593
                            self.loop_closing = True
                        if d2car < min_distance:                          # :598
                            self.current_cone_index = j; min_distance = d2car
                j += 1
            if d2car < self.map_thr and not found and not self.loop_closing:      # :608
                self._add_cone(gxy[i], int(ty), zxy[i])
            if self.loop_closing:                                         # :625
                if self.quirks:
                    self._optimise(); self.loop_closing_complete = True   # once per remaining observation (§8-B.2)
                else:
                    pending = True
        if pending:
            self._optimise(); self.loop_closing_complete = True

    def _loop_closing_candidate(self, c, d2car):                          # loopClosing, :697-706
        d = np.sqrt((self.map[0][0] - c[0]) ** 2 + (self.map[0][1] - c[1]) ** 2)
        return d < 1 and self.current_cone_index > 20 and d2car < self.map_thr

    # localizer, :340-414
    def _localizer(self, pose, cones):
        K = len(cones)
        zxy = self.fe.polar_to_xy(cones[:, 0], cones[:, 1], cones[:, 2])
        gxy = self.fe.cone_to_global(pose[None], np.zeros(K, dtype=np.int32), cones)
        reobserved, min_distance, current = 0, 100.0, None                # currentConeIndex is uninitialised at :348
        for i in range(K):
            d2car, ty = cones[i, 2], int(cones[i, 3])                     # static_cast<int>(coneObservedGlobal(2)), :357
            j, found = 0, False
            while not found and j < len(self.map):                        # :355
                c = self.map[j]
                dt = c[2] - ty                                            # int - int, no fabs in the reference (:360)
                type_ok = (dt < 0.0001) if self.quirks else (abs(dt) < 0.0001)
                if np.sqrt((c[0] - gxy[i, 0]) ** 2 + (c[1] - gxy[i, 1]) ** 2) < self.thr and type_ok:
                    reobserved += 1; found = True
                    # addConeMeasurement(m_map[j], pose), :373: the POSE goes where (az, zen, dist) is expected (§8-B.4)
                    z = self.fe.polar_to_xy([pose[0]], [pose[1]], [pose[2]])[0] if self.quirks else zxy[i]
                    self._add_measurement(c[3], z)
                    if d2car < min_distance:                              # :375-378
                        current = j; min_distance = d2car
                j += 1
        if reobserved > 0:                                                # :387
            self.current_cone_index = current
        self.send_pose = self.g.poses()[self.n_poses - 1].copy()          # updatePoseFromGraph, :404-408, :416-422
        self.localizer_calls += 1


class RefShell:
    """Test-only restatement of the reference's process shell over RefSlam: main()'s senderStamp filters and seven triggers
    (reference src/opendlv-logic-cfsd18-sensation-slam.cpp:65-108), Slam::setUp's keys (src/slam.cpp:736-756), the message
    intake nextSplitPose / nextPose / nextYawRate / nextCone (:67-219), the gathering wait + keyframe gate
    (initializeCollection :221-257, isKeyframe :286-295; the detached busy-wait thread becomes poll(now)), and what the
    localizer publishes (sendPose + sendCones, :404-410, 656-695).  Geodesy = the reference's own header
    (oracle/_ref/libref_wgs84.so) and cone messages = the reference's own Cone class (oracle/_ref/libref_cone.so) when
    built.  The checker for csrc/gs_shell.cpp + gs_shell_cluon.*: nothing here calls the product."""
    WGS84, ANGULAR_VELOCITY, HEADING, GEOLOCATION, OBJECT_TYPE, OBJECT_DIRECTION, OBJECT_DISTANCE = 19, 1031, 1051, 1116, 1131, 1133, 1134
    PI = float(np.float32(3.14159265))                                    # src/slam.hpp:136: a float literal

    def __init__(self, argv, quirks=False):
        a = {}
        for s in argv[1:]:
            if s.startswith("--"):
                k, _, v = s[2:].partition("="); a[k] = v if v else "1"
        self.gathering_ms = int(a["gatheringTimeMs"]); self.tbk = float(a["timeBetweenKeyframes"])       # setUp, :739-745
        self.cones_per_packet = int(a["conesPerPacket"]); self.sender_stamp = int(a["id"])
        self.gps_ref = (float(a["refLatitude"]), float(a["refLongitude"]))
        self.detect, self.estimation = int(a["detectConeId"]), int(a["estimationId"])                    # main, :67-68
        self.quirks = bool(quirks)
        self.slam = RefSlam(same_cone_threshold=float(a["sameConeThreshold"]), cone_mapping_threshold=float(a["coneMappingThreshold"]), quirks=quirks)
        self.odometry = np.zeros(3)                                       # m_odometryData
        self.geolocation_us = 0; self.yaw_us = 0; self.last_cone_us = 0   # m_geolocationReceivedTime, m_yawReceivedTime, m_lastTimeStamp
        self.frame_open = False; self.frame_opened_us = 0; self.keyframe_us = 0      # m_keyframeTimeStamp() = 0
        self.out = []; self.frames_run = 0; self.frames_gated = 0

    def _to_cartesian(self, lat, lon):
        if po.ref_wgs84() is None:
            raise RuntimeError("oracle/_ref/libref_wgs84.so not built")
        return po.ref_to_cartesian(self.gps_ref, (lat, lon))

    def on_message(self, data_type, sender_stamp, sample_us, now_us, object_id=0, v=(0.0, 0.0, 0.0)):
        S = self.slam
        if data_type in (self.WGS84, self.HEADING, self.GEOLOCATION, self.ANGULAR_VELOCITY):
            if sender_stamp != self.estimation:                           # :71-100: envelope.senderStamp() == senderStamp
                return 0
            if data_type == self.WGS84:                                   # nextSplitPose, :156-176
                self.odometry[:2] = self._to_cartesian(v[0], v[1])
            elif data_type == self.HEADING:                               # :177-184
                h = v[0] - self.PI
                h = h - 2 * self.PI if h > self.PI else h
                h = h + 2 * self.PI if h < -self.PI else h
                self.odometry[2] = h
            elif data_type == self.GEOLOCATION:                           # nextPose, :187-209
                self.geolocation_us = int(sample_us)
                xy = self._to_cartesian(v[0], v[1]); self.odometry[:] = (xy[0], xy[1], v[2])
            else:                                                         # nextYawRate, :211-219
                S.next_yaw_rate(v[0]); self.yaw_us = int(sample_us)
            return 1
        if data_type in (self.OBJECT_DIRECTION, self.OBJECT_DISTANCE, self.OBJECT_TYPE):
            if sender_stamp != self.detect:
                return 0
            self.last_cone_us = int(sample_us)                            # m_lastTimeStamp = data.sampleTimeStamp(), :73,102,129
            if data_type == self.OBJECT_DIRECTION: opened = S.collect_direction(int(object_id), v[0], v[1])
            elif data_type == self.OBJECT_DISTANCE: opened = S.collect_distance(int(object_id), v[0])
            else: opened = S.collect_type(int(object_id), int(v[0]))
            if opened:                                                    # std::thread coneCollector(&Slam::initializeCollection, this), :94-95
                self.frame_open = True; self.frame_opened_us = int(now_us)
            return 1
        return 0                                                          # no trigger registered for this type, :102-108

    def poll(self, now_us):
        """initializeCollection once its busy-wait has ended (elapsed > m_timeDiffMilliseconds * 1000, :227-233)."""
        S = self.slam
        if not self.frame_open or now_us - self.frame_opened_us <= self.gathering_ms * 1000:
            return 0
        self.frame_open = False
        extracted = S.collector[:, :S.last_object_id + 1].T.copy()        # leftCols(m_lastObjectId + 1), :241
        S.new_frame = True; S.last_object_id = 0; S.collector[:] = 0.0    # :242-244
        if extracted.shape[0] == 0:                                       # :248
            return 0
        elapsed_ms = abs(float(now_us - self.keyframe_us)) / 1000         # isKeyframe, :286-295
        if not elapsed_ms > self.tbk:
            self.frames_gated += 1; return 0
        self.keyframe_us = int(now_us)
        S.set_sample_times(self.yaw_us, self.last_cone_us)
        calls = S.localizer_calls
        S.perform(self.odometry.copy(), extracted)                        # performSLAM, :298-338 (its own 200 m guard first)
        self.frames_run += 1
        if S.localizer_calls > calls:                                     # sendPose(); sendCones(); at the end of localizer, :409-410
            self._send_pose(); self._send_cones()
        return 1

    def _send_pose(self):                                                 # :679-695
        S = self.slam
        gps = po.ref_from_cartesian(self.gps_ref, S.send_pose[:2])       # {latitude, longitude}
        lon_field = np.float32(gps[0]) if self.quirks else np.float32(gps[1])       # poseMessage.longitude(sendGPS[0]): the reference's swap (§8-B.7)
        lat_field = np.float32(gps[1]) if self.quirks else np.float32(gps[0])
        self.out.append((self.GEOLOCATION, self.sender_stamp, self.geolocation_us, 0, (float(lat_field), float(lon_field), float(np.float32(S.send_pose[2])))))

    def _send_cones(self):                                                # :656-677
        S = self.slam; n = len(S.map)
        for i in range(self.cones_per_packet):
            idx = S.current_cone_index + i
            idx = idx if idx < n else idx - n                             # the reference's single wrap, :666-667
            c = S.map[idx % n]
            if po.ref_cone() is not None and self.quirks:                 # the reference's own Cone::getDirection / getDistance
                az, _, di = po.ref_cone_encode(c[0], c[1], c[2], c[3], S.send_pose)
            else:
                x, y = c[0] - S.send_pose[0], c[1] - S.send_pose[1]
                heading = S.send_pose[2] * (1 / 57.295779513082325) if self.quirks else S.send_pose[2] * 57.295779513082325
                az = np.float32(np.arctan2(y, x) * 57.295779513082325 - heading); di = np.float32(np.sqrt(x * x + y * y))
            st = (self.sender_stamp, self.geolocation_us, i)
            self.out.append((self.OBJECT_DIRECTION,) + st + ((float(az), 0.0, 0.0),))
            self.out.append((self.OBJECT_DISTANCE,) + st + ((float(di), 0.0, 0.0),))
            self.out.append((self.OBJECT_TYPE,) + st + ((float(c[2]), 0.0, 0.0),))

    def take_output(self):
        o, self.out = self.out, []
        return o
