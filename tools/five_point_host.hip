// Debug harness: runs the GPU library's five-point solver source on the host
// CPU (no GPU needed) so it can be compared with the oracle during development.
// Not part of libvo_hip.so.
#include "../visual_odometry_amd/csrc/vo_ransac.hip"

extern "C" int host_five_point(const double* x1, const double* x2, double* Es) {
    static double sA[200 * vo::ST];
    return vo::five_point_dev(x1, x2, Es, sA);
}
extern "C" int host_five_point_chart(const double* x1, const double* x2, double* Es, int chart) {
    static double sA[200 * vo::ST];
    return vo::five_point_dev(x1, x2, Es, sA, chart);
}
extern "C" int host_front(const double* x1, const double* x2, int chart, double* cf, double* bm, double* EE) {
    static double sA[200 * vo::ST];
    return vo::five_point_front(x1, x2, sA, (double (*)[9])EE, (double (*)[13])bm, cf, chart) ? 1 : 0;
}
extern "C" int host_real_roots(const double* c, double* roots) { return vo::real_roots(c, roots); }
