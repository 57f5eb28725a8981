/*
 * ref_wgs84.cpp — the reference's OWN WGS84 <-> Cartesian transforms, compiled from the header where it lies
 * (-I/root/reference/src, reference src/WGS84toCartesian.hpp:39-146, nothing copied).  TEST INFRASTRUCTURE: output
 * goes to oracle/_ref/libref_wgs84.so only (git-ignored, travels to the GPU box).  It is the checker for the product's
 * csrc/gs_geo.cpp (row f-4 of SURVEY 8f: odometry intake, reference src/slam.cpp:154-219, and sendPose :679-695).
 */
#include <array>
#include <cmath>
#include <cstdint>
#include <limits>
#include "WGS84toCartesian.hpp"

extern "C" {
void ref_wgs84_to_cartesian(const double ref_latlon[2], const double pos_latlon[2], double out_xy[2]) {
    const std::array<double, 2> r{ref_latlon[0], ref_latlon[1]}, p{pos_latlon[0], pos_latlon[1]};
    const std::array<double, 2> o = wgs84::toCartesian(r, p);
    out_xy[0] = o[0]; out_xy[1] = o[1];
}
void ref_wgs84_from_cartesian(const double ref_latlon[2], const double xy[2], double out_latlon[2]) {
    const std::array<double, 2> r{ref_latlon[0], ref_latlon[1]}, c{xy[0], xy[1]};
    const std::array<double, 2> o = wgs84::fromCartesian(r, c);
    out_latlon[0] = o[0]; out_latlon[1] = o[1];
}
}
