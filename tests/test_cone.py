"""Row f-2 (output encoders), CPU only: the product's cone encoder (gs_cone_encode, csrc/gs_slam.cpp — what
gs_slam_encode_cones / the shell's sendCones call per cone) against the REFERENCE'S OWN Cone class
(src/cone.cpp:34-53 getDirection / getDistance, :55-85 accessors) compiled where it lies into oracle/_ref/libref_cone.so
(recipe: oracle/Makefile ref_shell) — a pinned oracle (kind "reference"), like libref_wgs84.so for row f-4."""
import numpy as np
import pytest

RAD2DEG = 57.295779513082325          # reference src/cone.hpp:55


@pytest.fixture(scope="module")
def ref(po):
    if po.ref_cone() is None:
        pytest.skip("oracle/_ref/libref_cone.so not built (needs /root/reference at build time: make -C oracle ref_shell)")
    return po


def cases(seed, n):
    rng = np.random.default_rng(seed)
    for _ in range(n):
        pose = np.array([rng.uniform(-150, 150), rng.uniform(-150, 150), rng.uniform(-np.pi, np.pi)])
        cone = pose[:2] + rng.uniform(-70, 70, 2)
        yield cone, pose
    # on the axes, behind the car, at the car, kilometre coordinates
    for cone, pose in (((5.0, 0.0), (0.0, 0.0, 0.0)), ((-5.0, 0.0), (0.0, 0.0, 0.0)), ((0.0, -3.0), (0.0, 0.0, 1.0)),
                       ((2.0, 2.0), (2.0, 2.0, 0.3)), ((4000.5, -3999.25), (4010.0, -3990.0, -3.0)), ((-5.0, -0.0), (0.0, 0.0, 0.0))):
        yield np.array(cone), np.array(pose)


def test_reference_quirk_mode_is_the_reference_bit_for_bit(pkg, ref):
    """cfg.reference_quirks = 1 is the reference's arithmetic as written (heading * 1 / RAD2DEG subtracted from degrees,
    src/cone.cpp:37-39): same operations in the same order => identical float32 fields."""
    for cone, pose in cases(11, 4000):
        az, di = pkg.cone_encode(cone, pose, reference_quirks=1)
        raz, rzen, rdi = ref.ref_cone_encode(cone[0], cone[1], 1, 0, pose)
        assert rzen == 0.0
        assert az.tobytes() == raz.tobytes() and di.tobytes() == rdi.tobytes(), (cone, pose, az, raz, di, rdi)


def test_clean_mode_is_the_reference_formula_with_the_heading_in_degrees(pkg, ref):
    """cfg.reference_quirks = 0 differs from the reference in ONE factor: the heading is converted with RAD2DEG instead of
    1 / RAD2DEG.  Feeding the reference's own code the heading theta * RAD2DEG^2 makes it evaluate exactly that (its
    1 / RAD2DEG cancels one factor), so the clean mode is pinned by the same binary: float32 fields within one ulp (the
    extra multiply / divide pair rounds twice), distance identical."""
    worst = 0.0
    for cone, pose in cases(12, 4000):
        az, di = pkg.cone_encode(cone, pose, reference_quirks=0)
        p2 = pose.copy(); p2[2] = pose[2] * RAD2DEG * RAD2DEG
        raz, _, rdi = ref.ref_cone_encode(cone[0], cone[1], 2, 7, p2)
        assert di.tobytes() == rdi.tobytes()
        worst = max(worst, abs(float(az) - float(raz)) / max(np.spacing(np.float32(abs(raz))), np.float32(1e-30)))
    assert worst <= 1.0, worst


def test_record_accessors_of_the_reference_cone(ref):
    """Cone is a plain {x, y, type, id} record (src/cone.hpp:51-54, src/cone.cpp:55-85): what MapCone mirrors."""
    out = np.zeros(4)
    ref.ref_cone().ref_cone_record(1.5, -2.5, 1, 3, 7.25, 8.5, 4, 9, out.ctypes.data_as(__import__("ctypes").POINTER(__import__("ctypes").c_double)))
    assert out.tolist() == [7.25, 8.5, 4.0, 9.0]


def test_null_arguments_are_refused(pkg):
    import ctypes as C
    L = pkg.binding.lib()
    az, di = C.c_float(), C.c_float()
    assert L.gs_cone_encode(None, None, 0, C.byref(az), C.byref(di)) < 0
