/*
 * ref_cone.cpp — the reference's OWN Cone class (map element + message encoders), compiled from the sources where they
 * lie: /root/reference/src/cone.cpp + cone.hpp (-I/root/reference/src), the vendored Eigen (-I/root/reference/thirdparty)
 * and the message set the reference's cluon-msc generates from its .odvd (oracle/_ref/gen, recipe `ref_shell`).  Nothing
 * is copied.  TEST INFRASTRUCTURE: output goes to oracle/_ref/libref_cone.so only (git-ignored, travels to the GPU box).
 * It is the checker for the product's cone encoders (gs_cone_encode / gs_slam_encode_cones, csrc/gs_slam.cpp), row f-2 of
 * Cone::getDirection / Cone::getDistance, reference src/cone.cpp:34-53, as Slam::sendCones calls them
 * (reference src/slam.cpp:656-677).
 */
#include "cone.hpp"

extern "C" {
/* Cone(x, y, type, id).getDirection(pose) / .getDistance(pose): the float32 message fields */
void ref_cone_encode(double x, double y, int type, int id, const double pose[3], float *azimuth, float *zenith, float *distance) {
    Cone c(x, y, type, id);
    Eigen::Vector3d p(pose[0], pose[1], pose[2]);
    opendlv::logic::perception::ObjectDirection d = c.getDirection(p);
    opendlv::logic::perception::ObjectDistance r = c.getDistance(p);
    *azimuth = d.azimuthAngle(); *zenith = d.zenithAngle(); *distance = r.distance();
}
/* accessors / mutators of the record (src/cone.cpp:55-85): out = {x, y, type, id} after set*(...) */
void ref_cone_record(double x, double y, int type, int id, double nx, double ny, int ntype, int nid, double out[4]) {
    Cone c(x, y, type, id);
    c.setX(nx); c.setY(ny); c.setType(ntype); c.setId(nid);
    out[0] = c.getX(); out[1] = c.getY(); out[2] = c.getType(); out[3] = c.getId();
}
}
