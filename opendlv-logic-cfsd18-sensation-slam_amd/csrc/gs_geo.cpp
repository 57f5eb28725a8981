// gs_geo.cpp — WGS84 <-> local Cartesian, host side (row f-4 of SURVEY 8f).
//
// What the reference does with a GPS fix before it becomes an odometry pose (Slam::nextSplitPose / nextPose,
// reference src/slam.cpp:154-209, through wgs84::toCartesian, src/WGS84toCartesian.hpp:39-113) and with the pose it
// sends back (Slam::sendPose, src/slam.cpp:679-695, through wgs84::fromCartesian, :119-146).
//
// The projection of that header is the ellipsoidal American polyconic projection about the reference point (Snyder,
// "Map Projections - A Working Manual", USGS PP 1395, eqs. 18-12 / 18-13), with the meridional distance evaluated by
// the five-term series in e^2 that PROJ uses (pj_enfn / pj_mlfn); its inverse is not the analytic one but a search:
// walk the latitude, then the longitude, in steps of 1e-5 degree while the Cartesian miss keeps shrinking and is above
// 1e-2 m.  Both are restated here from those descriptions; tests/test_geo.py pins them against the reference's own
// header compiled into oracle/_ref/libref_wgs84.so.
#include "graphslam.h"
#include <cmath>
#include <limits>

namespace {
constexpr double kPi = 3.141592653589793;
constexpr double kDegToRad = kPi / 180.0, kHalfPi = kPi / 2.0;
constexpr double kEquatorRadius = 6378137.0, kFlattening = 1.0 / 298.257223563;
constexpr double kEs = 2.0 * kFlattening - kFlattening * kFlattening;          // first eccentricity squared

struct MeridianSeries {                                                        // PROJ pj_enfn: coefficients of the meridional distance
    double en[5];
    MeridianSeries() {
        const double es = kEs;
        en[0] = 1.0 - es * (0.25 + es * (0.046875 + es * (0.01953125 + es * 0.01068115234375)));
        en[1] = es * (0.75 - es * (0.046875 + es * (0.01953125 + es * 0.01068115234375)));
        const double es2 = es * es;
        en[2] = es2 * (0.46875 - es * (0.01302083333333333333 + es * 0.00712076822916666666));
        const double es3 = es2 * es;
        en[3] = es3 * (0.36458333333333333333 - es * 0.00569661458333333333);
        en[4] = es3 * es * 0.3076171875;
    }
    double distance(double phi) const {                                        // pj_mlfn, in units of the equatorial radius
        const double s = std::sin(phi), sc = std::cos(phi) * s, s2 = s * s;
        return en[0] * phi - sc * (en[1] + s2 * (en[2] + s2 * (en[3] + s2 * en[4])));
    }
};

void polyconic_forward(const double ref_deg[2], const double pos_deg[2], double out_xy[2]) {
    static const MeridianSeries M;
    double lat = pos_deg[0] * kDegToRad, lon = pos_deg[1] * kDegToRad;
    const double pole_gap = std::fabs(lat) - kHalfPi;
    out_xy[0] = 0.0; out_xy[1] = 0.0;
    if (pole_gap > 1.0e-12 || std::fabs(lon) > 10.0) return;                   // beyond a pole / not a longitude: the reference returns the origin
    if (std::fabs(pole_gap) < 1.0e-12) lat = lat < 0.0 ? -kHalfPi : kHalfPi;
    const double m0 = M.distance(ref_deg[0] * kDegToRad);
    const double dlon = lon - ref_deg[1] * kDegToRad;
    double x = dlon, y = -m0;                                                  // on the equator the parallels are straight
    if (!(std::fabs(lat) < 1.0e-10)) {
        const double s = std::sin(lat);
        const double ms = std::fabs(s) > 1.0e-10 ? (std::cos(lat) / std::sqrt(1.0 - kEs * s * s)) / s : 0.0;   // N cot(phi) / a
        const double e = dlon * s;
        x = ms * std::sin(e);
        y = (M.distance(lat) - m0) + ms * (1.0 - std::cos(e));
    }
    out_xy[0] = kEquatorRadius * x; out_xy[1] = kEquatorRadius * y;
}
}  // namespace

extern "C" int gs_wgs84_to_cartesian(const double ref_latlon_deg[2], const double pos_latlon_deg[2], double out_xy[2]) {
    if (!ref_latlon_deg || !pos_latlon_deg || !out_xy) return GS_ERR_INVALID;
    polyconic_forward(ref_latlon_deg, pos_latlon_deg, out_xy);
    return GS_OK;
}

extern "C" int gs_wgs84_from_cartesian(const double ref_latlon_deg[2], const double xy[2], double out_latlon_deg[2]) {
    if (!ref_latlon_deg || !xy || !out_latlon_deg) return GS_ERR_INVALID;
    const double stop = 1.0e-2, step = 1.0e-5;                                 // metres, degrees
    const double lat_step = (xy[1] < 0 ? -1 : 1) * step, lon_step = (xy[0] < 0 ? -1 : 1) * step;
    double guess[2] = {ref_latlon_deg[0], ref_latlon_deg[1]}, c[2];
    polyconic_forward(ref_latlon_deg, guess, c);
    double before = std::numeric_limits<double>::max(), miss = std::fabs(xy[1] - c[1]);
    while (miss < before && miss > stop) {                                     // northing first (the last step taken is the one that stopped improving)
        guess[0] += lat_step; polyconic_forward(ref_latlon_deg, guess, c);
        before = miss; miss = std::fabs(xy[1] - c[1]);
    }
    before = std::numeric_limits<double>::max(); miss = std::fabs(xy[0] - c[0]);
    while (miss < before && miss > stop) {                                     // then easting
        guess[1] += lon_step; polyconic_forward(ref_latlon_deg, guess, c);
        before = miss; miss = std::fabs(xy[0] - c[0]);
    }
    out_latlon_deg[0] = guess[0]; out_latlon_deg[1] = guess[1];
    return GS_OK;
}
