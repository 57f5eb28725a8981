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


def close_f32(a, b, ulps=1):
    """float32 message fields computed from estimates that agree to ~1e-10 (GPU vs CPU oracle): equal, or `ulps` apart when the
    double lands next to a rounding boundary"""
    a, b = np.float32(a), np.float32(b)
    return a == b or abs(float(a) - float(b)) <= ulps * float(np.spacing(np.float32(max(abs(a), abs(b)))))


def same_messages(got, want, S):
    """What the shell published against what the restated reference shell (tests/ref_slam.py RefShell, over the CPU oracle,
    the reference's own geodesy header and Cone class) publishes for the same input stream."""
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g[:4] == w[:4], (g, w)                                # message id, --id sender stamp, sample time, objectId
        if g[0] == S.GEOLOCATION:
            assert close_f32(g[4][0], w[4][0]) and close_f32(g[4][1], w[4][1]) and abs(g[4][2] - w[4][2]) <= 1e-6, (g, w)
        elif g[0] == S.OBJECT_DIRECTION:
            assert abs(g[4][0] - w[4][0]) <= 2e-5 and g[4][1] == 0 and w[4][1] == 0, (g, w)      # degrees, float32
        elif g[0] == S.OBJECT_DISTANCE:
            assert abs(g[4][0] - w[4][0]) <= 2e-6 * max(1.0, abs(w[4][0])), (g, w)
        else:
            assert g[4][0] == w[4][0], (g, w)


@pytest.mark.gpu
@pytest.mark.parametrize("quirks", [0, 1])
def test_shell_replays_a_lap_and_publishes_what_the_restated_reference_shell_publishes(pkg, po, quirks):
    """A lap of the synthetic track as the message stream the microservice sees (Geolocation + yaw rate per frame, the three
    cone messages per object in scrambled order, messages of foreign senders in between, one frame inside the keyframe period,
    one frame whose odometry is beyond 200 m), through gs_shell_on_message / gs_shell_poll (product: csrc/gs_shell.cpp over the
    HIP C-ABI) AND through tests/ref_slam.py RefShell — main() + the timing glue of Slam restated from the reference's text
    over the CPU oracle, the reference's own WGS84 header and Cone class (oracle/_ref).  Every return value, the counters, the
    map and every published message must agree: before the loop closes nothing is published; from the closing frame on every
    frame publishes one Geolocation and conesPerPacket x (direction, distance, type), stamped --id and the sample time of the
    last Geolocation message (src/slam.cpp:656-695)."""
    from ref_slam import RefShell
    if po.ref_wgs84() is None:
        pytest.skip("oracle/_ref/libref_wgs84.so not built")
    N, M = 120, 60
    t = pkg.track.generate(N, M)
    argv = list(ARGV) + (["--referenceQuirks"] if quirks else [])
    argv = [a if not a.startswith("--coneMappingThreshold") else "--coneMappingThreshold=67" for a in argv]
    argv = [a if not a.startswith("--conesPerPacket") else "--conesPerPacket=5" for a in argv]
    S = pkg.Shell(argv)
    R = RefShell(argv, quirks=bool(quirks))
    ref = (57.70924648, 11.9462)
    rng = np.random.default_rng(7)
    now = 1_000_000
    published = 0
    def both(*a, **kw):
        x, y = S.on_message(*a, **kw), R.on_message(*a, **kw)
        assert x == y, (a, kw, x, y)
        return x
    frames = list(range(N)) + list(range(8))
    for n, k in enumerate(frames):
        gated = n == 40                                         # a frame 0.3 s after the previous keyframe: extracted, dropped (isKeyframe)
        far = n == 50                                           # odometry 250 m off: performSLAM returns at once (:300-303)
        now += 300_000 if gated else 600_000                    # otherwise 0.6 s between frames: every frame is a keyframe (500 ms)
        pose = t["odom_poses"][k].copy(); obs = np.asarray(t["obs"][k])
        if far:
            pose[0] += 250.0
        lat, lon = po.ref_from_cartesian(ref, pose[:2]) if not far else (ref[0] + 0.00225, ref[1])       # ~250 m north of the reference point
        sample = 50_000_000 + 100_000 * n
        wz = float(np.float32(rng.normal(0, 0.2)))
        assert both(S.GEOLOCATION, 112, sample, now, v=(lat, lon, pose[2])) == 1
        assert both(S.GEOLOCATION, 7, sample, now, v=(0.0, 0.0, 9.0)) == 0            # a foreign sender
        assert both(S.ANGULAR_VELOCITY, 112, sample + 30_000, now, v=(wz, 0, 0)) == 1
        if n == 3:                                              # the split-pose triggers (GeodeticWgs84Reading + GeodeticHeadingReading), then the Geolocation again
            assert both(S.WGS84, 112, sample, now, v=(lat + 1e-5, lon, 0)) == 1 and both(S.HEADING, 112, sample, now, v=(4.0, 0, 0)) == 1
            assert both(S.HEADING, 116, sample, now, v=(1.0, 0, 0)) == 0
            assert both(S.GEOLOCATION, 112, sample, now, v=(lat, lon, pose[2])) == 1
        msgs = [(f, i) for i in range(len(obs)) for f in range(3)]
        for q in rng.permutation(len(msgs)):
            f, i = msgs[q]
            ty, v = ((S.OBJECT_DIRECTION, (np.float32(obs[i, 0]), np.float32(obs[i, 1]), 0)), (S.OBJECT_DISTANCE, (np.float32(obs[i, 2]), 0, 0)),
                     (S.OBJECT_TYPE, (obs[i, 3], 0, 0)))[f]
            assert both(ty, 116, sample + 10_000, now + 100, object_id=i, v=v) == 1
            assert both(ty, 3, sample, now + 100, object_id=i + 1, v=(1, 1, 1)) == 0                # a foreign sender's cones
        assert S.poll(now + 100 + 19_000) == 0 and R.poll(now + 100 + 19_000) == 0      # inside the gathering window
        a, b = S.poll(now + 100 + 20_001), R.poll(now + 100 + 20_001)
        assert a == b == (0 if gated else 1)
        assert S.counters() == (R.frames_run, R.frames_gated)
        assert S.slam.map_size == len(R.slam.map) and S.slam.loop_closed == R.slam.loop_closing_complete
        assert S.slam.current_cone_index == R.slam.current_cone_index
        assert S.slam.graph.n_poses == R.slam.n_poses           # the far frame added no pose on either side
        got, want = S.take_output(), R.take_output()
        same_messages(got, want, S)
        if got:
            published += 1
            assert len(got) == 1 + 3 * 5 and all(o[1] == 120 and o[2] == sample for o in got)      # --id, m_geolocationReceivedTime
            assert R.slam.loop_closing_complete
    assert published >= 8 and S.counters() == (N + 8 - 1, 1)
    xs, ts = S.slam.map()
    assert np.array_equal(ts, [c[2] for c in R.slam.map]) and np.allclose(xs, [[c[0], c[1]] for c in R.slam.map], rtol=0, atol=1e-7)
    S.close()
    with pytest.raises(pkg.GsError):
        S.slam.map_size                                         # the borrowed handle went with the shell (no use-after-free)


@pytest.mark.gpu
def test_cluon_binding_in_process_with_encoded_envelopes(pkg, po):
    """csrc/gs_shell_cluon.hpp — the binding of the shell to libcluon — compiled against the reference's own cluon header and
    the message set its generator makes from the reference's .odvd (recipe: oracle/Makefile ref_shell, outputs in the
    git-ignored oracle/_ref/), driven by tests/shell_cluon_driver.cpp: every message of a lap is serialised to the OD4 wire
    format, parsed back and handed to the triggers; what the shell publishes goes through the typed send path into
    Envelopes, is decoded again and printed.  Checked against tests/ref_slam.py RefShell (the reference's shell restated over
    the CPU oracle, the reference's geodesy header and Cone class) fed the same decoded stream."""
    import os
    import subprocess
    from ref_slam import RefShell
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "oracle", "_ref", "shell_cluon_driver")
    if not os.path.exists(drv) or po.ref_wgs84() is None:
        pytest.skip("oracle/_ref/shell_cluon_driver not built (needs /root/reference: make -C oracle ref_shell)")
    argv = [a if not a.startswith("--coneMappingThreshold") else "--coneMappingThreshold=67" for a in ARGV]
    argv = [a if not a.startswith("--conesPerPacket") else "--conesPerPacket=5" for a in argv]
    r = subprocess.run([drv] + argv[1:], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    got = [(int(a[0]), int(a[1]), int(a[2]), int(a[3]), (float(a[4]), float(a[5]), float(a[6]))) for a in (l.split() for l in lines)]
    tail = [l for l in r.stdout.splitlines() if l.startswith("#")][0]
    # the same stream, decoded, through the restated reference shell
    N, M = 120, 60
    t = pkg.track.generate(N, M); ref = (57.70924648, 11.9462)
    S = RefShell(argv, quirks=False)
    now, want = 1_000_000, []
    for n in range(N + 8):
        k = n if n < N else n - N
        now += 600_000; sample = 50_000_000 + 100_000 * n
        pose = t["odom_poses"][k]; obs = t["obs"][k]
        lat, lon = pkg.wgs84_from_cartesian(ref, pose[:2])       # what the driver puts on the wire (doubles)
        S.on_message(S.GEOLOCATION, 112, sample, now, v=(lat, lon, np.float32(pose[2]))); S.on_message(S.GEOLOCATION, 7, sample, now, v=(lat, lon, np.float32(pose[2])))
        S.on_message(S.ANGULAR_VELOCITY, 112, sample + 30_000, now, v=(np.float32(0.01) * np.float32(n % 7), 0, 0))
        for i in range(len(obs)):
            S.on_message(S.OBJECT_TYPE, 116, sample + 10_000, now + 100, object_id=i, v=(int(obs[i, 3]), 0, 0))
            S.on_message(S.OBJECT_DIRECTION, 116, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 0]), np.float32(obs[i, 1]), 0))
            S.on_message(S.OBJECT_DISTANCE, 116, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 2]), 0, 0))
            S.on_message(S.OBJECT_DISTANCE, 3, sample + 10_000, now + 100, object_id=i, v=(np.float32(obs[i, 2]), 0, 0))
        assert S.poll(now + 100 + 20_001) == 1
        want += S.take_output()
    assert S.slam.loop_closing_complete and len(want) > 0
    assert tail == "# frames run %d gated 0 map %d loop_closed 1" % (N + 8, len(S.slam.map))
    same_messages(got, want, S)
