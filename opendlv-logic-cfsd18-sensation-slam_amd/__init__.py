"""MI355X-native GraphSLAM back-end for the cone-landmark optimiser of
cfsd/opendlv-logic-cfsd18-sensation-slam (reference src/slam.cpp / src/cone.cpp).

The product is csrc/libgraphslam_hip.so (hand-written HIP for gfx950 behind include/graphslam.h).
This package only holds the ctypes plumbing above that C-ABI and the synthetic track source.
The directory name contains hyphens, import it with
    importlib.import_module("opendlv-logic-cfsd18-sensation-slam_amd")
"""
from . import binding, track  # noqa: F401
from .binding import Config, Graph, GsError, Shell, Slam, Stats, cone_encode, default_config, device_count, wgs84_from_cartesian, wgs84_to_cartesian  # noqa: F401


def build(force=False):
    """Compile every native component of the package (HIP library for gfx950 + track generator)."""
    binding.build(force=force)
    track.build(force=force)
