/*
 * ref_eigen.cpp — the reference's OWN linear solver, built from the reference's vendored Eigen
 * 3.3.4 where it lies (-I/root/reference/thirdparty, nothing copied).  TEST INFRASTRUCTURE:
 * output goes to oracle/_ref/libref_eigen.so only (git-ignored, travels to the GPU box).
 *
 * It restates the ~20 lines of g2o's LinearSolverEigen::solve that the reference instantiates
 * (reference src/slam.cpp:55-59: BlockSolverX + LinearSolverEigen, setBlockOrdering(false)):
 *   first solve after initializeOptimization : analyzePattern  (scalar AMD ordering,
 *        reference thirdparty/Eigen/src/OrderingMethods/Ordering.h:52-81 -> Amd.h:94, and the
 *        elimination tree, thirdparty/Eigen/src/SparseCholesky/SimplicialCholesky_impl.h:51-98)
 *   every solve : factorize (…_impl.h:101-190), then x = chol.solve(b)
 *        (thirdparty/Eigen/src/SparseCholesky/SimplicialCholesky.h:421-494).
 * g2o (unpinned, mid-2018) derived its decomposition from SimplicialLDLT<SparseMatrix,Upper>;
 * later versions use SimplicialLLT.  Both are available here (kind 0 = LDLT, 1 = LLT).
 */
#include <Eigen/Sparse>
#include <Eigen/SparseCholesky>
#include <Eigen/Geometry>
#include <cstdio>
#include <chrono>
#include <cstring>

namespace {
typedef Eigen::SparseMatrix<double, Eigen::ColMajor> SpMat;
struct RefSolver {
    int kind;
    Eigen::SimplicialLDLT<SpMat, Eigen::Upper> ldlt;
    Eigen::SimplicialLLT<SpMat, Eigen::Upper> llt;
    SpMat A;
    double ms_analyze, ms_factor, ms_solve;
    long calls;
};
double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
}

extern "C" {

void *ref_eigen_create(int kind) {
    RefSolver *s = new RefSolver();
    s->kind = kind; s->ms_analyze = s->ms_factor = s->ms_solve = 0; s->calls = 0;
    return s;
}
void ref_eigen_destroy(void *ctx) { delete static_cast<RefSolver *>(ctx); }

/* signature == orc_solver_fn */
int ref_eigen_solve(void *ctx, int analyze, int n, const int *colptr, const int *rowind,
                    const double *values, const double *b, double *x) {
    RefSolver *s = static_cast<RefSolver *>(ctx);
    const int nnz = colptr[n];
    if (analyze || s->A.rows() != n || s->A.nonZeros() != nnz) {
        s->A.resize(n, n);
        s->A.resizeNonZeros(nnz);
        std::memcpy(s->A.outerIndexPtr(), colptr, sizeof(int) * (size_t)(n + 1));
        std::memcpy(s->A.innerIndexPtr(), rowind, sizeof(int) * (size_t)nnz);
        analyze = 1;
    }
    std::memcpy(s->A.valuePtr(), values, sizeof(double) * (size_t)nnz);
    auto t0 = std::chrono::steady_clock::now();
    if (analyze) { if (s->kind == 0) s->ldlt.analyzePattern(s->A); else s->llt.analyzePattern(s->A); }
    s->ms_analyze += analyze ? ms_since(t0) : 0.0;
    t0 = std::chrono::steady_clock::now();
    bool ok;
    if (s->kind == 0) { s->ldlt.factorize(s->A); ok = s->ldlt.info() == Eigen::Success; }
    else { s->llt.factorize(s->A); ok = s->llt.info() == Eigen::Success; }
    s->ms_factor += ms_since(t0);
    if (!ok) return -1;
    t0 = std::chrono::steady_clock::now();
    Eigen::Map<const Eigen::VectorXd> bb(b, n);
    Eigen::Map<Eigen::VectorXd> xx(x, n);
    if (s->kind == 0) xx = s->ldlt.solve(bb); else xx = s->llt.solve(bb);
    s->ms_solve += ms_since(t0);
    s->calls++;
    return 0;
}

void ref_eigen_timings(void *ctx, double out[3]) {
    RefSolver *s = static_cast<RefSolver *>(ctx);
    out[0] = s->ms_analyze; out[1] = s->ms_factor; out[2] = s->ms_solve;
}
void ref_eigen_reset_timings(void *ctx) {
    RefSolver *s = static_cast<RefSolver *>(ctx);
    s->ms_analyze = s->ms_factor = s->ms_solve = 0; s->calls = 0;
}
const char *ref_eigen_version(void) {
    static char buf[64];
    snprintf(buf, sizeof buf, "Eigen %d.%d.%d", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION);
    return buf;
}
/* Rotation2D sanity hook used by a test: R(theta) as the reference's Eigen builds it
 * (thirdparty/Eigen/src/Geometry/Rotation2D.h:188) */
void ref_eigen_rotation2d(double theta, double out[4]) {
    Eigen::Matrix2d R = Eigen::Rotation2Dd(theta).toRotationMatrix();
    out[0] = R(0, 0); out[1] = R(0, 1); out[2] = R(1, 0); out[3] = R(1, 1);
}
}
