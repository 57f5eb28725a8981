"""Row f-4 (odometry intake, pose output), CPU only: the product's host-side WGS84 <-> Cartesian transforms
(csrc/gs_geo.cpp) against the REFERENCE'S OWN header (src/WGS84toCartesian.hpp:39-146) compiled where it lies into
oracle/_ref/libref_wgs84.so — a pinned oracle (kind "reference"), unlike the g2o arithmetic."""
import numpy as np
import pytest


@pytest.fixture(scope="module")
def ref(po):
    if po.ref_wgs84() is None:
        pytest.skip("oracle/_ref/libref_wgs84.so not built (needs /root/reference at build time)")
    return po


REFS = [(57.719264, 11.957199), (48.7823, 9.1770), (-33.8568, 151.2153), (0.0, 0.0), (64.1, -21.9)]   # Gothenburg first: where the reference team drove


def test_to_cartesian_matches_the_reference_header(pkg, ref):
    rng = np.random.default_rng(3)
    worst = 0.0
    for r in REFS:
        for _ in range(400):
            pos = np.array(r) + rng.uniform(-0.02, 0.02, 2)            # within ~2 km of the reference point
            a = pkg.wgs84_to_cartesian(r, pos); b = ref.ref_to_cartesian(r, pos)
            worst = max(worst, np.abs(a - b).max())
    assert worst < 1e-8                                               # metres; same series, same order of evaluation up to constant folding
    # the reference's guards: on the equator, at a pole, and a "longitude" that is not one
    for r, pos in (((10.0, 20.0), (0.0, 20.5)), ((80.0, 0.0), (90.0, 3.0)), ((10.0, 20.0), (95.0, 20.0)), ((10.0, 20.0), (10.0, 700.0))):
        assert np.allclose(pkg.wgs84_to_cartesian(r, pos), ref.ref_to_cartesian(r, pos), rtol=0, atol=1e-8)


def test_from_cartesian_matches_the_reference_step_search(pkg, ref):
    """The reference inverts by walking 1e-5 degree steps (src/WGS84toCartesian.hpp:119-146): the answer is quantised and
    overshoots by one step by construction; the product must land on the same grid point."""
    rng = np.random.default_rng(4)
    for r in REFS[:3]:
        for _ in range(25):
            xy = rng.uniform(-300, 300, 2)
            a = pkg.wgs84_from_cartesian(r, xy); b = ref.ref_from_cartesian(r, xy)
            assert np.abs(a - b).max() < 1e-9, (r, xy, a, b)
            back = pkg.wgs84_to_cartesian(r, a)
            assert np.abs(back - xy).max() < 2.5                       # within a couple of steps of the target (1e-5 deg ~ 1.1 m)


def test_slam_intake_and_pose_encoding(pkg, ref):
    """Slam::nextSplitPose / nextPose / nextYawRate / sendPose (reference src/slam.cpp:154-219, 679-695) on a host-only
    handle: heading wrap with the float PI, yaw rate / 4, latitude / longitude swap of sendPose under the quirk flag."""
    r = REFS[0]
    PI = float(np.float32(3.14159265))
    for quirks in (0, 1):
        S = pkg.Slam(device=-2, reference_quirks=quirks)
        S.set_gps_reference(*r)
        S.next_wgs84(r[0] + 0.001, r[1] - 0.002)
        for nh in (0.3, 3.5, 6.2, -0.2):
            S.next_heading(nh)
            h = nh - PI; h = h - 2 * PI if h > PI else h; h = h + 2 * PI if h < -PI else h
            assert S.odometry()[2] == h
        assert np.allclose(S.odometry()[:2], ref.ref_to_cartesian(r, (r[0] + 0.001, r[1] - 0.002)), rtol=0, atol=1e-8)
        S.next_geolocation(r[0] - 0.0005, r[1] + 0.0007, 1.25); S.next_yaw_rate(0.8)
        o = S.odometry()
        assert np.allclose(o[:2], ref.ref_to_cartesian(r, (r[0] - 0.0005, r[1] + 0.0007)), rtol=0, atol=1e-8) and o[2] == 1.25 and o[3] == float(np.float32(0.8) / np.float32(4))    # float m_yawRate, src/slam.hpp:128
        enc = S.encode_pose()                                          # send pose is still (0, 0, 0): the reference point itself
        latlon = ref.ref_from_cartesian(r, (0.0, 0.0))
        want = (latlon[0], latlon[1]) if quirks else (latlon[1], latlon[0])
        assert enc[0] == np.float32(want[0]) and enc[1] == np.float32(want[1]) and enc[2] == np.float32(0.0)
        S.close()
