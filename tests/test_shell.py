"""The microservice shell: csrc/gs_shell.cpp against what main() of the reference does
(src/opendlv-logic-cfsd18-sensation-slam.cpp:49-119): command line, senderStamp filters of the seven triggers,
gathering window, keyframe gate, published messages.  The dispatch tests run on a host-only handle (no GPU)."""
import numpy as np
import pytest

ARGV = ["opendlv-logic-cfsd18-sensation-slam", "--cid=111", "--id=120", "--detectConeId=116", "--estimationId=112",
        "--gatheringTimeMs=20", "--sameConeThreshold=1.2", "--refLatitude=57.70924648", "--refLongitude=11.9462",
        "--timeBetweenKeyframes=500", "--coneMappingThreshold=50", "--conesPerPacket=20"]       # usecase/docker-compose.yml:16


def test_command_line_like_the_reference(pkg):
    with pytest.raises(pkg.GsError) as e:                       # fewer than 10 arguments: usage, exit code 1 (:52-57)
        pkg.Shell(ARGV[:6], device=-2)
    assert "Usage" in str(e.value) and "--cid=" in str(e.value)
    with pytest.raises(pkg.GsError):                            # enough arguments, but a key std::stoi would throw on
        pkg.Shell([a for a in ARGV if not a.startswith("--conesPerPacket")] + ["--verbose", "--x=1"], device=-2)
    S = pkg.Shell(ARGV + ["--verbose"], device=-2)
    assert S.cid == 111
    S.close()


def test_sender_stamp_filters_and_odometry_intake(pkg):
    S = pkg.Shell(ARGV, device=-2)
    ref = (57.70924648, 11.9462)
    # estimation messages: taken with --estimationId (112) only (:71-100)
    assert S.on_message(S.WGS84, 999, 1, 1, v=(ref[0] + 0.001, ref[1] - 0.002, 0)) == 0
    assert np.array_equal(S.slam.odometry(), np.zeros(4))
    assert S.on_message(S.WGS84, 112, 1, 1, v=(ref[0] + 0.001, ref[1] - 0.002, 0)) == 1
    assert np.allclose(S.slam.odometry()[:2], pkg.wgs84_to_cartesian(ref, (ref[0] + 0.001, ref[1] - 0.002)), rtol=0, atol=1e-12)
    assert S.on_message(S.HEADING, 112, 2, 2, v=(3.5, 0, 0)) == 1 and S.slam.odometry()[2] != 0
    assert S.on_message(S.GEOLOCATION, 112, 3, 3, v=(ref[0], ref[1], 1.25)) == 1 and S.slam.odometry()[2] == 1.25
    assert S.on_message(S.ANGULAR_VELOCITY, 116, 4, 4, v=(0.8, 0, 0)) == 0          # the cone stamp is not the estimation stamp
    assert S.on_message(S.ANGULAR_VELOCITY, 112, 4, 4, v=(0.8, 0, 0)) == 1
    assert S.slam.odometry()[3] == float(np.float32(0.8) / np.float32(4))
    # cone messages: --detectConeId (116) only; an unknown message type is ignored
    assert S.on_message(S.OBJECT_DIRECTION, 112, 5, 5, object_id=0, v=(10.0, 0, 0)) == 0
    assert S.on_message(12345, 116, 5, 5) == 0
    assert S.poll(10_000_000) == 0                              # nothing collected: no frame is open
    assert S.on_message(S.OBJECT_DIRECTION, 116, 5, 1000, object_id=0, v=(10.0, 0, 0)) == 1
    assert S.on_message(S.OBJECT_DISTANCE, 116, 5, 1500, object_id=0, v=(7.0, 0, 0)) == 1
    assert S.on_message(S.OBJECT_TYPE, 116, 5, 1600, object_id=0, v=(1, 0, 0)) == 1
    assert S.poll(1000 + 20_000) == 0                           # the gathering window (20 ms since the frame's FIRST message) has not passed
    # keyframe gate: |now - m_keyframeTimeStamp| in ms must exceed --timeBetweenKeyframes (m_keyframeTimeStamp starts at 0):
    # at now = 21.001 ms the frame is extracted, found to be no keyframe and dropped (src/slam.cpp:245-252, 286-295)
    assert S.poll(1000 + 20_001) == 0
    assert S.counters() == (0, 1)
    assert S.poll(10_000_000) == 0                              # the dropped frame is gone: the collector was reset
    S.close()


@pytest.mark.gpu
@pytest.mark.parametrize("quirks", [0, 1])
def test_shell_replays_a_lap_like_the_slam_mirror_and_publishes_after_loop_closure(pkg, quirks):
    """A lap of the synthetic track as the message stream the microservice sees (Geolocation + yaw rate per frame, the three
    cone messages per object in scrambled order, messages of foreign senders in between), through gs_shell_on_message /
    gs_shell_poll, against the Slam mirror driven directly with the same frames.  Before the loop closes nothing is
    published; from the closing frame on every frame publishes one Geolocation and conesPerPacket x (direction,
    distance, type), stamped --id and the sample time of the last Geolocation message (src/slam.cpp:656-695)."""
    N, M = 120, 60
    t = pkg.track.generate(N, M)
    argv = list(ARGV) + (["--referenceQuirks"] if quirks else [])
    argv = [a if not a.startswith("--coneMappingThreshold") else "--coneMappingThreshold=67" for a in argv]
    argv = [a if not a.startswith("--conesPerPacket") else "--conesPerPacket=5" for a in argv]
    S = pkg.Shell(argv)
    D = pkg.Slam(same_cone_threshold=1.2, cone_mapping_threshold=67.0, reference_quirks=quirks)
    ref = (57.70924648, 11.9462); D.set_gps_reference(*ref)
    rng = np.random.default_rng(7)
    now = 1_000_000
    published = 0
    for n, k in enumerate(list(range(N)) + list(range(8))):
        now += 600_000                                          # 0.6 s between frames: every frame is a keyframe (500 ms)
        pose = t["odom_poses"][k]; obs = np.asarray(t["obs"][k])
        lat, lon = pkg.wgs84_from_cartesian(ref, pose[:2])       # the odometry as the Geolocation message carries it
        sample = 50_000_000 + 100_000 * n
        wz = float(np.float32(rng.normal(0, 0.2)))
        for X in ("shell", "direct"):
            if X == "shell":
                assert S.on_message(S.GEOLOCATION, 112, sample, now, v=(lat, lon, pose[2])) == 1
                assert S.on_message(S.GEOLOCATION, 7, sample, now, v=(0.0, 0.0, 9.0)) == 0     # a foreign sender
                assert S.on_message(S.ANGULAR_VELOCITY, 112, sample + 30_000, now, v=(wz, 0, 0)) == 1
            else:
                D.next_geolocation(lat, lon, pose[2]); D.next_yaw_rate(wz); D.set_sample_times(sample + 30_000, sample + 10_000)
        msgs = [(f, i) for i in range(len(obs)) for f in range(3)]
        for q in rng.permutation(len(msgs)):
            f, i = msgs[q]
            ty, v = ((S.OBJECT_DIRECTION, (np.float32(obs[i, 0]), np.float32(obs[i, 1]), 0)), (S.OBJECT_DISTANCE, (np.float32(obs[i, 2]), 0, 0)),
                     (S.OBJECT_TYPE, (obs[i, 3], 0, 0)))[f]
            assert S.on_message(ty, 116, sample + 10_000, now + 100, object_id=i, v=v) == 1
            assert S.on_message(ty, 3, sample, now + 100, object_id=i + 1, v=(1, 1, 1)) == 0                 # a foreign sender's cones
        assert S.poll(now + 100 + 19_000) == 0                  # inside the gathering window
        assert S.poll(now + 100 + 20_001) == 1
        obs32 = obs.copy(); obs32[:, :3] = obs[:, :3].astype(np.float32)    # the message fields are float32
        D.perform_slam(D.odometry()[:3], obs32)
        assert S.slam.map_size == D.map_size and S.slam.loop_closed == D.loop_closed and S.slam.current_cone_index == D.current_cone_index
        out = S.take_output()
        if not D.loop_closed:
            assert out == []
        else:
            published += 1
            assert len(out) == 1 + 3 * 5
            assert all(o[1] == 120 and o[2] == sample for o in out)                 # --id, m_geolocationReceivedTime
            ep = D.encode_pose(); az, di, ty = D.encode_cones(5)
            assert out[0][0] == S.GEOLOCATION and np.allclose(out[0][4], (ep[1], ep[0], ep[2]), rtol=0, atol=0)
            for i in range(5):
                a, b, c = out[1 + 3 * i: 4 + 3 * i]
                assert (a[0], b[0], c[0]) == (S.OBJECT_DIRECTION, S.OBJECT_DISTANCE, S.OBJECT_TYPE) and a[3] == b[3] == c[3] == i
                assert a[4][0] == az[i] and a[4][1] == 0 and b[4][0] == di[i] and c[4][0] == ty[i]
    assert published >= 8 and S.counters()[0] == N + 8
    xs, ts = S.slam.map(); xd, td = D.map()
    assert np.array_equal(ts, td) and np.array_equal(xs, xd)
    S.close(); D.close()


@pytest.mark.gpu
def test_cluon_binding_in_process_with_encoded_envelopes(pkg):
    """csrc/gs_shell_cluon.hpp — the binding of the shell to libcluon — compiled against the reference's own cluon header and
    the message set its generator makes from the reference's .odvd (recipe: oracle/Makefile ref_shell, outputs in the
    git-ignored oracle/_ref/), driven by tests/shell_cluon_driver.cpp: every message of a lap is serialised to the OD4 wire
    format, parsed back and handed to the triggers; what the shell publishes goes through the typed send path into
    Envelopes, is decoded again and printed.  The Python shell fed the same decoded stream must publish the same values."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "oracle", "_ref", "shell_cluon_driver")
    if not os.path.exists(drv):
        pytest.skip("oracle/_ref/shell_cluon_driver not built (needs /root/reference: make -C oracle ref_shell)")
    argv = [a if not a.startswith("--coneMappingThreshold") else "--coneMappingThreshold=67" for a in ARGV]
    argv = [a if not a.startswith("--conesPerPacket") else "--conesPerPacket=5" for a in argv]
    r = subprocess.run([drv] + argv[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    got = [(int(a[0]), int(a[1]), int(a[2]), int(a[3]), (float(a[4]), float(a[5]), float(a[6]))) for a in (l.split() for l in lines)]
    tail = [l for l in r.stdout.splitlines() if l.startswith("#")][0]
    # the same stream through the Python binding of the same shell
    N, M = 120, 60
    t = pkg.track.generate(N, M); ref = (57.70924648, 11.9462)
    S = pkg.Shell(argv)
    now, want = 1_000_000, []
    for n in range(N + 8):
        k = n if n < N else n - N
        now += 600_000; sample = 50_000_000 + 100_000 * n
        pose = t["odom_poses"][k]; obs = t["obs"][k]
        lat, lon = pkg.wgs84_from_cartesian(ref, pose[:2])
        S.on_message(S.GEOLOCATION, 112, sample, now, v=(lat, lon, np.float32(pose[2]))); S.on_message(S.GEOLOCATION, 7, sample, now, v=(lat, lon, np.float32(pose[2])))
        S.on_message(S.ANGULAR_VELOCITY, 112, sample + 30_000, now, v=(np.float32(0.01) * np.float32(n % 7), 0, 0))
        for i in range(len(obs)):
            S.on_message(S.OBJECT_TYPE, 116, sample + 10_000, now + 100, object_id=i, v=(int(obs[i, 3]), 0, 0))
            S.on_message(S.OBJECT_DIRECTION, 116, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 0]), np.float32(obs[i, 1]), 0))
            S.on_message(S.OBJECT_DISTANCE, 116, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 2]), 0, 0))
            S.on_message(S.OBJECT_DISTANCE, 3, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 2]), 0, 0))
        assert S.poll(now + 100 + 20_001) == 1
        want += S.take_output()
    assert S.slam.loop_closed and len(want) > 0
    assert tail == "# frames run %d gated 0 map %d loop_closed 1" % (N + 8, S.slam.map_size)
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:4] == w[:4]
        if g[0] == S.GEOLOCATION:       # latitude / longitude are doubles on the wire, the heading a float
            assert g[4][0] == w[4][0] and g[4][1] == w[4][1] and g[4][2] == float(np.float32(w[4][2]))
        else:                           # float32 fields (the shell's values are float32 already) / integer type
            assert g[4] == tuple(float(np.float32(x)) for x in w[4])
    S.close()
