// Essential-matrix RANSAC and pose recovery on gfx950.  Stands in for
// cv::findEssentialMat(RANSAC) + cv::recoverPose as called by
// ref:src/motion_estimator.cpp:69-85.
//
// Hypotheses are generated in rounds: the index sets come from the same
// multiply-with-carry stream cv::RNG(-1) produces, a round of 5-point solves
// runs one hypothesis per lane (fp64, 10x20 elimination matrix in LDS), every
// (hypothesis, model) is scored by a wavefront over the points (Sampson error,
// ballot-free per-lane counts reduced across the wave), and one lane per pair
// then replays the sequential RANSAC bookkeeping (strictly-greater best model,
// adaptive iteration count) over the round's counts, so the winner is the one
// the serial loop would have picked.
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>

#include "vo_internal.h"

namespace vo {

constexpr int kMaxModels = 10;
constexpr int kMaxIters = 1000;

struct RState {
    unsigned long long rng;
    int n;
    int niters;
    int iters_run;
    int max_good;
    int done;
    int exact5;
    int models_scored;   // candidates of the hypotheses the serial loop has walked
    double bestE[9];
};

struct RParams {
    double ifx, ify, bx, by;   // normalisation x*ifx + bx
    double fx_mean;
    double threshold;          // already divided by (fx+fy)/2
    float t2;                  // (float)(threshold^2)
    double t2_mid;             // midpoint between t2 and the next float above it
    double confidence;
    int fixed_iters;
};

// ------------------------------------------------------------------ points
__global__ void normalize_kernel(const float* __restrict__ p1, const float* __restrict__ p2,
                                 const int32_t* __restrict__ n_arr, int n_max, RParams P, double* __restrict__ m,
                                 float* __restrict__ mf) {
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    int n = n_arr ? min(n_arr[b], n_max) : n_max;
    if (i >= n) return;
    size_t o = (size_t)b * n_max + i;
    double x1 = (double)p1[o * 2] * P.ifx + P.bx, y1 = (double)p1[o * 2 + 1] * P.ify + P.by;
    double x2 = (double)p2[o * 2] * P.ifx + P.bx, y2 = (double)p2[o * 2 + 1] * P.ify + P.by;
    double* d = m + o * 4;
    d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
    // single-precision copy for the scoring kernel's first look at a point (half the bytes)
    *reinterpret_cast<float4*>(mf + o * 4) = make_float4((float)x1, (float)y1, (float)x2, (float)y2);
}

__device__ __forceinline__ uint32_t rng_next(unsigned long long& s) {
    s = (unsigned long long)(uint32_t)s * 4164903690u + (uint32_t)(s >> 32);
    return (uint32_t)s;
}

// Draws `count` 5-subsets (distinct indices, rejection as in getSubset).
__device__ void draw_samples(RState& st, int32_t* out, int count) {
    unsigned long long s = st.rng;
    uint32_t n = (uint32_t)st.n;
    for (int h = 0; h < count; ++h) {
        int idx[5];
        for (int i = 0; i < 5; ++i) {
            int v;
            bool dup;
            do {
                v = (int)(rng_next(s) % n);
                dup = false;
                for (int k = 0; k < i; ++k) dup |= idx[k] == v;
            } while (dup);
            idx[i] = v;
            out[h * 5 + i] = v;
        }
    }
    st.rng = s;
}

__device__ int update_num_iters(double p, double ep, int max_iters) {
    p = fmin(fmax(p, 0.), 1.);
    ep = fmin(fmax(ep, 0.), 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, 5.0);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : __double2int_rn(num / denom);
}

__global__ void init_kernel(RState* __restrict__ st, const int32_t* __restrict__ n_arr, int n_max, int batch,
                            RParams P, int32_t* __restrict__ samples, int samples_stride, int first_round,
                            int32_t* __restrict__ retry_list) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) retry_list[0] = 0;
    if (b >= batch) return;
    RState s;
    s.rng = ~0ull;
    s.n = n_arr ? min(n_arr[b], n_max) : n_max;
    s.niters = P.fixed_iters > 0 ? P.fixed_iters : kMaxIters;
    s.iters_run = 0;
    s.max_good = 0;
    s.done = s.n < 5;
    s.exact5 = s.n == 5;
    s.models_scored = 0;
    for (int k = 0; k < 9; ++k) s.bestE[k] = 0;
    int32_t* out = samples + (size_t)b * samples_stride;
    if (s.exact5) {
        s.niters = 1;
        for (int i = 0; i < 5; ++i) out[i] = i;
    } else if (!s.done) {
        draw_samples(s, out, min(first_round, s.niters));
    }
    st[b] = s;
}

// ------------------------------------------------------------------ five-point solver
// From here to the scoring kernel multiplications and additions may fuse: every model the solver
// hands on is the fixed point of the refinement on the constraints, which does not depend on how
// the elimination and the root search before it were rounded, and the fused forms halve the
// dependent chains these kernels consist of.  (Scoring, bookkeeping and pose recovery keep the
// translation unit's default - no contraction -, where an operation order is part of the result.)
#pragma clang fp contract(fast)
constexpr int ST = 32;   // lanes per solve workgroup (one hypothesis each): 51 KB of LDS, so three workgroups
                         // share a CU, or one fits beside the resident tiles of an image kernel.  Sixteen lanes
                         // (25 KB) left twice as many quarter-filled wavefronts issuing and measured 3 % slower
                         // on the solve kernels; forty-eight measured no better than thirty-two.

// quad[10] (+)= s * a[4] (x) b[4]
__host__ __device__ __forceinline__ void mul_ll(const double* a, const double* b, double s, double* q) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int QI[4][4] = {{0, 1, 2, 3}, {1, 4, 5, 6}, {2, 5, 7, 8}, {3, 6, 8, 9}};
            q[QI[i][j]] += s * (a[i] * b[j]);
        }
}

// row of A (LDS, stride ST doubles) += s * q[10] (x) l[4]
__host__ __device__ __forceinline__ void mul_ql_row(const double* q, const double* l, double s, double* row) {
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int CI[10][4] = {{0, 2, 4, 5},    {2, 3, 8, 9},     {4, 8, 10, 11},   {5, 9, 11, 12},
                                       {3, 1, 6, 7},    {8, 6, 13, 14},   {9, 7, 14, 15},   {10, 13, 16, 17},
                                       {11, 14, 17, 18}, {12, 15, 18, 19}};
            row[CI[i][j] * ST] += s * (q[i] * l[j]);
        }
}

// row[20] (registers) += s * q[10] (x) l[4]
__host__ __device__ __forceinline__ void mul_ql_reg(const double* q, const double* l, double s, double* row) {
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            constexpr int CI[10][4] = {{0, 2, 4, 5},    {2, 3, 8, 9},     {4, 8, 10, 11},   {5, 9, 11, 12},
                                       {3, 1, 6, 7},    {8, 6, 13, 14},   {9, 7, 14, 15},   {10, 13, 16, 17},
                                       {11, 14, 17, 18}, {12, 15, 18, 19}};
            row[CI[i][j]] += s * (q[i] * l[j]);
        }
}

// j-th derivative of the polynomial c (degree n) at x; dc[j] holds its coefficients.
__host__ __device__ __forceinline__ double poly_eval(const double* d, int deg, double x) {
    double v = d[deg];
    for (int i = deg - 1; i >= 0; --i) v = v * x + d[i];
    return v;
}

// Real roots (ascending) of c[0..10] by derivative interlacing: the roots of
// each derivative bracket the monotone pieces of the next lower one, and
// every bracket with a sign change is closed by a Newton/bisection hybrid.
// Extrema of the polynomial itself that vanish to rounding level are reported
// as (double) roots; roots closer than 1e-7 relative are reported once.
__host__ __device__ int real_roots(const double* c, double* roots) {
    double cmax = 0;
    for (int i = 0; i <= 10; ++i) cmax = fmax(cmax, fabs(c[i]));
    if (!(cmax > 0) || !(cmax < DBL_MAX)) return 0;
    int n = 10;
    while (n > 0 && fabs(c[n]) <= 1e-14 * cmax) --n;
    if (n == 0) return 0;
    double dc[11][11];   // dc[j] = coefficients of the j-th derivative
    for (int i = 0; i <= n; ++i) dc[0][i] = c[i];
    for (int j = 1; j < n; ++j)
        for (int i = 0; i <= n - j; ++i) dc[j][i] = dc[j - 1][i + 1] * (double)(i + 1);
    double prev[10], cur[10];
    int np = 0;
    {
        const double* d = dc[n - 1];   // linear
        prev[0] = -d[0] / d[1];
        np = 1;
    }
    for (int j = n - 2; j >= 0; --j) {
        const double* d = dc[j];
        const double* dd = dc[j + 1];
        int deg = n - j;
        double R = 0;
        for (int i = 0; i < deg; ++i) R = fmax(R, fabs(d[i] / d[deg]));
        R += 1.0;
        int nc = 0;
        double a = -R, fa = poly_eval(d, deg, a);
        bool pending = false;   // the extremum at `a` touches zero and the piece left of it held no root
        for (int k = 0; k <= np; ++k) {
            double b = k < np ? fmin(fmax(prev[k], -R), R) : R;
            double fb = poly_eval(d, deg, b);
            const bool changed = b > a && fa != 0.0 && (fa < 0) != (fb < 0) && fb != 0.0;
            if (pending && b > a) {
                // no root on either side of the extremum: it is the double root itself
                bool dup = nc > 0 && fabs(a - cur[nc - 1]) <= 1e-7 * fmax(1.0, fabs(a));
                if (!changed && !dup && nc < 10) cur[nc++] = a;
                pending = false;
            }
            if (b > a) {
                if (fa == 0.0) {
                    if (nc == 0 || cur[nc - 1] != a) cur[nc++] = a;
                } else if (changed) {
                    // bracketed root: safeguarded Newton (Numerical-Recipes style rtsafe)
                    double xl = fa < 0 ? a : b, xh = fa < 0 ? b : a;
                    double x = 0.5 * (a + b), dxold = fabs(b - a), dx = dxold;
                    double f = poly_eval(d, deg, x), df = poly_eval(dd, deg - 1, x);
                    for (int it = 0; it < 100; ++it) {
                        if ((((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
                            dxold = dx;
                            dx = 0.5 * (xh - xl);
                            x = xl + dx;
                        } else {
                            dxold = dx;
                            dx = f / df;
                            x -= dx;
                        }
                        if (fabs(dx) <= 4.0 * DBL_EPSILON * fmax(1.0, fabs(x))) break;
                        f = poly_eval(d, deg, x);
                        df = poly_eval(dd, deg - 1, x);
                        if (f == 0.0) break;
                        if (f < 0.0) xl = x; else xh = x;
                    }
                    bool dup = j == 0 && nc > 0 && fabs(x - cur[nc - 1]) <= 1e-7 * fmax(1.0, fabs(x));
                    if (nc < 10 && !dup) cur[nc++] = x;
                }
            }
            if (j == 0 && k < np && fb != 0.0 && b > a) {
                // A local extremum that comes within 1e-4 of zero (relative to the polynomial's
                // magnitude there) with no sign change on either side is kept as a candidate once
                // the next piece is known to hold no root either: a double root (exactly degenerate
                // data, e.g. pure sideways translation) or a pair of close roots that the rounding
                // of the elimination has merged away; the refinement on the constraints decides.
                // (Next to a sign change the bracketed roots already are the two solutions.)
                double mag = fabs(d[deg]), ab = fabs(b);
                for (int i = deg - 1; i >= 0; --i) mag = mag * ab + fabs(d[i]);
                pending = !changed && fabs(fb) <= 1e-4 * mag;
            }
            if (b > a) {
                a = b;
                fa = fb;
            }
        }
        if (fa == 0.0 && nc < 10 && (nc == 0 || cur[nc - 1] != a)) cur[nc++] = a;
        np = nc;
        for (int k = 0; k < nc; ++k) prev[k] = cur[k];
    }
    for (int k = 0; k < np; ++k) roots[k] = prev[k];
    return np;
}

// ---- degree-10 fast path of real_roots(): the same derivative-interlacing scheme with
// every polynomial held in registers (statically unrolled per degree); only the two
// short root lists live in this lane's LDS column.
__host__ __device__ constexpr double falling(int m, int j) {   // m! / (m-j)!
    double r = 1.0;
    for (int i = 0; i < j; ++i) r *= (double)(m - i);
    return r;
}

template <int DEG>
__host__ __device__ __forceinline__ double horner(const double (&d)[DEG + 1], double x) {
    double v = d[DEG];
#pragma unroll
    for (int i = DEG - 1; i >= 0; --i) v = v * x + d[i];
    return v;
}

// Roots of the (10-DEG)-th derivative of c (degree DEG) given the np roots of the next
// derivative in sP (ascending).  Writes them ascending to sC, returns their number.
template <int DEG>
__host__ __device__ __forceinline__ int roots_level(const double* c, const double* sP, int np, double* sC) {
    constexpr int J = 10 - DEG;
    double d[DEG + 1], dd[DEG];
#pragma unroll
    for (int i = 0; i <= DEG; ++i) d[i] = c[i + J] * falling(i + J, J);
#pragma unroll
    for (int i = 0; i < DEG; ++i) dd[i] = d[i + 1] * (double)(i + 1);
    double R = 0;
#pragma unroll
    for (int i = 0; i < DEG; ++i) R = fmax(R, fabs(d[i] / d[DEG]));
    R += 1.0;
    int nc = 0;
    double last = 0;   // last root written (valid when nc > 0)
    double a = -R, fa = horner<DEG>(d, a);
    bool pending = false;   // the extremum at `a` touches zero and the piece left of it held no root
    for (int k = 0; k <= np; ++k) {
        double b = k < np ? fmin(fmax(sP[k * ST], -R), R) : R;
        double fb = horner<DEG>(d, b);
        const bool changed = b > a && fa != 0.0 && (fa < 0) != (fb < 0) && fb != 0.0;
        if (DEG == 10 && pending && b > a) {
            bool dup = nc > 0 && fabs(a - last) <= 1e-7 * fmax(1.0, fabs(a));
            if (!changed && !dup && nc < 10) { sC[nc * ST] = a; last = a; ++nc; }
            pending = false;
        }
        if (b > a) {
            if (fa == 0.0) {
                if (nc == 0 || last != a) { sC[nc * ST] = a; last = a; ++nc; }
            } else if (changed) {
                double xl = fa < 0 ? a : b, xh = fa < 0 ? b : a;
                double x = 0.5 * (a + b), dxold = fabs(b - a), dx = dxold;
                double f = horner<DEG>(d, x), df = horner<DEG - 1>(dd, x);
                for (int it = 0; it < 100; ++it) {
                    if ((((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
                        dxold = dx;
                        dx = 0.5 * (xh - xl);
                        x = xl + dx;
                    } else {
                        dxold = dx;
                        dx = f / df;
                        x -= dx;
                    }
                    if (fabs(dx) <= 4.0 * DBL_EPSILON * fmax(1.0, fabs(x))) break;
                    f = horner<DEG>(d, x);
                    df = horner<DEG - 1>(dd, x);
                    if (f == 0.0) break;
                    if (f < 0.0) xl = x; else xh = x;
                }
                bool dup = DEG == 10 && nc > 0 && fabs(x - last) <= 1e-7 * fmax(1.0, fabs(x));
                if (nc < 10 && !dup) { sC[nc * ST] = x; last = x; ++nc; }
            }
        }
        if (DEG == 10 && k < np && fb != 0.0 && b > a) {
            double ab = fabs(b);
            double mag = fabs(d[DEG]);
#pragma unroll
            for (int i = DEG - 1; i >= 0; --i) mag = mag * ab + fabs(d[i]);
            pending = !changed && fabs(fb) <= 1e-4 * mag;
        }
        if (b > a) {
            a = b;
            fa = fb;
        }
    }
    if (fa == 0.0 && nc < 10 && (nc == 0 || last != a)) { sC[nc * ST] = a; ++nc; }
    return nc;
}

// c: degree-10 polynomial with a non-negligible leading coefficient.  Roots (ascending)
// end up in sW[0..n) (stride ST); sW needs 20 slots.
__host__ __device__ __forceinline__ int real_roots10(const double* c, double* sW) {
    double* A = sW;
    double* B = sW + 10 * ST;
    A[0] = -(c[9] * falling(9, 9)) / (c[10] * falling(10, 9));   // root of the 9th derivative
    int n = 1;
    n = roots_level<2>(c, A, n, B);
    n = roots_level<3>(c, B, n, A);
    n = roots_level<4>(c, A, n, B);
    n = roots_level<5>(c, B, n, A);
    n = roots_level<6>(c, A, n, B);
    n = roots_level<7>(c, B, n, A);
    n = roots_level<8>(c, A, n, B);
    n = roots_level<9>(c, B, n, A);
    n = roots_level<10>(c, A, n, B);
    for (int k = 0; k < n; ++k) A[k * ST] = B[k * ST];
    return n;
}

// The ten cubic constraints on an essential matrix, f(E) = [det E ;
// 2 E E^T E - trace(E E^T) E], and (dir != nullptr) their derivative along D.
__host__ __device__ __forceinline__ void ess_constraints(const double* E, double* f) {
    double G[9];   // E E^T
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) G[i * 3 + j] = E[i * 3] * E[j * 3] + E[i * 3 + 1] * E[j * 3 + 1] + E[i * 3 + 2] * E[j * 3 + 2];
    double tr = G[0] + G[4] + G[8];
    f[0] = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            f[1 + i * 3 + j] = 2.0 * (G[i * 3] * E[j] + G[i * 3 + 1] * E[3 + j] + G[i * 3 + 2] * E[6 + j]) - tr * E[i * 3 + j];
}

__host__ __device__ __forceinline__ void ess_constraints_dir(const double* E, const double* D, double* df) {
    const double C[9] = {E[4] * E[8] - E[5] * E[7], E[5] * E[6] - E[3] * E[8], E[3] * E[7] - E[4] * E[6],
                         E[2] * E[7] - E[1] * E[8], E[0] * E[8] - E[2] * E[6], E[1] * E[6] - E[0] * E[7],
                         E[1] * E[5] - E[2] * E[4], E[2] * E[3] - E[0] * E[5], E[0] * E[4] - E[1] * E[3]};
    double dd = 0, trEE = 0, trED = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        dd += C[k] * D[k];
        trEE += E[k] * E[k];
        trED += E[k] * D[k];
    }
    df[0] = dd;
    double EtE[9], EEt[9], DtE[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            EtE[i * 3 + j] = E[i] * E[j] + E[3 + i] * E[3 + j] + E[6 + i] * E[6 + j];
            EEt[i * 3 + j] = E[i * 3] * E[j * 3] + E[i * 3 + 1] * E[j * 3 + 1] + E[i * 3 + 2] * E[j * 3 + 2];
            DtE[i * 3 + j] = D[i] * E[j] + D[3 + i] * E[3 + j] + D[6 + i] * E[6 + j];
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                s += D[i * 3 + k] * EtE[k * 3 + j] + E[i * 3 + k] * DtE[k * 3 + j] + EEt[i * 3 + k] * D[k * 3 + j];
            df[1 + i * 3 + j] = 2.0 * s - 2.0 * trED * E[i * 3 + j] - trEE * D[i * 3 + j];
        }
}

// Gauss-Newton on (x,y,z) for E = x*B0 + y*B1 + z*B2 + B3 against the ten constraints, so
// that every model returned is the exact solution of the five-point problem to rounding
// whatever the conditioning of the elimination that produced the starting point.
// polish_steps takes at most `steps` steps and returns true once the iteration is over
// (converged: the last step moved nothing; or the normal equations are singular), so that
// it can be resumed: any split of the step budget gives the result of one uninterrupted run.
constexpr int kPolishSteps = 12;
__host__ __device__ bool polish_steps(const double (*B)[9], double& x, double& y, double& z, int steps) {
    for (int it = 0; it < steps; ++it) {
        double E[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = B[0][k] * x + B[1][k] * y + B[2][k] * z + B[3][k];
        double f[10], J0[10], J1[10], J2[10];
        ess_constraints(E, f);
        ess_constraints_dir(E, B[0], J0);
        ess_constraints_dir(E, B[1], J1);
        ess_constraints_dir(E, B[2], J2);
        double n00 = 0, n01 = 0, n02 = 0, n11 = 0, n12 = 0, n22 = 0, g0 = 0, g1 = 0, g2 = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            n00 += J0[k] * J0[k]; n01 += J0[k] * J1[k]; n02 += J0[k] * J2[k];
            n11 += J1[k] * J1[k]; n12 += J1[k] * J2[k]; n22 += J2[k] * J2[k];
            g0 += J0[k] * f[k]; g1 += J1[k] * f[k]; g2 += J2[k] * f[k];
        }
        double lam = 1e-14 * (n00 + n11 + n22);
        n00 += lam; n11 += lam; n22 += lam;
        double c00 = n11 * n22 - n12 * n12, c01 = n01 * n22 - n12 * n02, c02 = n01 * n12 - n11 * n02;
        double det = n00 * c00 - n01 * c01 + n02 * c02;
        if (!(fabs(det) > 0)) return true;
        double inv = 1.0 / det;
        double d0 = (g0 * c00 - n01 * (g1 * n22 - n12 * g2) + n02 * (g1 * n12 - n11 * g2)) * inv;
        double d1 = (n00 * (g1 * n22 - n12 * g2) - g0 * c01 + n02 * (n01 * g2 - g1 * n02)) * inv;
        double d2 = (n00 * (n11 * g2 - g1 * n12) - n01 * (n01 * g2 - g1 * n02) + g0 * c02) * inv;
        x -= d0; y -= d1; z -= d2;
        if (fmax(fabs(d0), fmax(fabs(d1), fabs(d2))) <= 1e-14 * fmax(1.0, fmax(fabs(x), fmax(fabs(y), fabs(z))))) return true;
    }
    return false;
}

// The unit-norm model of (x,y,z) in Eout; returns its largest constraint residual.
__host__ __device__ double polish_finish(const double (*B)[9], double x, double y, double z, double* Eout) {
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = B[0][k] * x + B[1][k] * y + B[2][k] * z + B[3][k];
    double nrm = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) nrm += E[k] * E[k];
    nrm = sqrt(nrm);
    if (!(nrm > 0) || !(nrm < DBL_MAX)) return 1e300;
#pragma unroll
    for (int k = 0; k < 9; ++k) Eout[k] = E[k] / nrm;
    double f[10], r = 0;
    ess_constraints(Eout, f);
#pragma unroll
    for (int k = 0; k < 10; ++k) r = fmax(r, fabs(f[k]));
    return r;
}

__host__ __device__ double polish_model_dev(const double (*B)[9], double x, double y, double z, double* Eout,
                                            int kSteps = kPolishSteps) {
    polish_steps(B, x, y, z, kSteps);
    return polish_finish(B, x, y, z, Eout);
}

// x1, x2: 5 normalised point pairs (x,y).  Writes up to 10 unit-norm
// essential matrices (row-major) to Es.  sA: this lane's column of the
// 10x20 elimination matrix in LDS (element (r,c) at sA[(r*20+c)*ST]).
__host__ __device__ bool five_point_front(const double* x1, const double* x2, double* sA, double (*EE)[9],
                                          double (*bm)[13], double* cf, int chart = 0) {
    // ---- null space of the 5x9 epipolar constraint matrix (Householder QR of its transpose)
    double M[9][5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        double a = x1[j * 2], b = x1[j * 2 + 1], c = x2[j * 2], d = x2[j * 2 + 1];
        M[0][j] = a * c; M[1][j] = b * c; M[2][j] = c;
        M[3][j] = a * d; M[4][j] = b * d; M[5][j] = d;
        M[6][j] = a; M[7][j] = b; M[8][j] = 1.0;
    }
    double beta[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        double s = 0;
#pragma unroll
        for (int i = k; i < 9; ++i) s += M[i][k] * M[i][k];
        double nrm = sqrt(s);
        double alpha = M[k][k] > 0 ? -nrm : nrm;
        double v0 = M[k][k] - alpha;
        double vn = s - M[k][k] * M[k][k] + v0 * v0;
        M[k][k] = v0;
        beta[k] = vn > 0 ? 2.0 / vn : 0.0;
#pragma unroll
        for (int j = k + 1; j < 5; ++j) {
            double dot = 0;
#pragma unroll
            for (int i = k; i < 9; ++i) dot += M[i][k] * M[i][j];
            dot *= beta[k];
#pragma unroll
            for (int i = k; i < 9; ++i) M[i][j] -= dot * M[i][k];
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        double e[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) e[i] = (i == 5 + t) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            double dot = 0;
#pragma unroll
            for (int i = k; i < 9; ++i) dot += M[i][k] * e[i];
            dot *= beta[k];
#pragma unroll
            for (int i = k; i < 9; ++i) e[i] -= dot * M[i][k];
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) EE[t][i] = e[i];
    }
    // The Householder basis inherits structure from the data (two equal rows of
    // the constraint matrix, as under pure sideways translation, put the true
    // solution exactly at infinity of the chart w = 1 used below).  Mixing the
    // four vectors with a fixed dense orthogonal matrix (I - 2uu^T,
    // u ~ (1, sqrt2, sqrt3, sqrt5)) removes such alignments.
    {
        // (chart 1: a second, unrelated mixing - the same solutions through a differently
        // conditioned elimination)
        const double u0[4] = {0.30151134457776363, 0.42640143271122088, 0.52223296786709361, 0.67419986246324204};
        const double u1[4] = {0.67419986246324204, -0.52223296786709361, 0.30151134457776363, -0.42640143271122088};
        const double u2[4] = {0.5, -0.5, -0.5, 0.5};
        const double* u = chart == 0 ? u0 : chart == 1 ? u1 : u2;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            double dot = u[0] * EE[0][i] + u[1] * EE[1][i] + u[2] * EE[2][i] + u[3] * EE[3][i];
#pragma unroll
            for (int t = 0; t < 4; ++t) EE[t][i] -= 2.0 * u[t] * dot;
        }
    }
    // ---- ten cubic constraints in (x,y,z) on E = x*EE0 + y*EE1 + z*EE2 + EE3
    double El[9][4];
#pragma unroll
    for (int rc = 0; rc < 9; ++rc)
#pragma unroll
        for (int t = 0; t < 4; ++t) El[rc][t] = EE[t][rc];
#define A_(r, c) sA[((r) * 20 + (c)) * ST]
    {
        // det(E) = 0
        double m0[10] = {0}, m1[10] = {0}, m2[10] = {0}, row[20];
#pragma unroll
        for (int i = 0; i < 20; ++i) row[i] = 0;
        mul_ll(El[4], El[8], 1.0, m0); mul_ll(El[5], El[7], -1.0, m0);
        mul_ll(El[3], El[8], 1.0, m1); mul_ll(El[5], El[6], -1.0, m1);
        mul_ll(El[3], El[7], 1.0, m2); mul_ll(El[4], El[6], -1.0, m2);
        mul_ql_reg(m0, El[0], 1.0, row);
        mul_ql_reg(m1, El[1], -1.0, row);
        mul_ql_reg(m2, El[2], 1.0, row);
#pragma unroll
        for (int i = 0; i < 20; ++i) A_(0, i) = row[i];
    }
    {
        // (E E^T - trace(E E^T)/2 I) E = 0
        double L[6][10];   // symmetric: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
        const int sr[6] = {0, 0, 0, 1, 1, 2}, sc[6] = {0, 1, 2, 1, 2, 2};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
            for (int i = 0; i < 10; ++i) L[q][i] = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) mul_ll(El[sr[q] * 3 + k], El[sc[q] * 3 + k], 1.0, L[q]);
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            double tr = 0.5 * (L[0][i] + L[3][i] + L[5][i]);
            L[0][i] -= tr; L[3][i] -= tr; L[5][i] -= tr;
        }
        const int sym[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double row[20];
#pragma unroll
                for (int i = 0; i < 20; ++i) row[i] = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) mul_ql_reg(L[sym[r][k]], El[k * 3 + c], 1.0, row);
#pragma unroll
                for (int i = 0; i < 20; ++i) A_(1 + r * 3 + c, i) = row[i];
            }
    }
    // ---- Gauss-Jordan on the first ten columns (partial pivoting), statically unrolled.
    // Only rows 4..9 are needed fully reduced, so rows 0..3 are never eliminated upwards.
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        int p = k;
        double best = fabs(A_(k, k));
#pragma unroll
        for (int r = k + 1; r < 10; ++r) {
            double v = fabs(A_(r, k));
            if (v > best) { best = v; p = r; }
        }
        if (!(best > 1e-300)) return false;
        double pr[20];
#pragma unroll
        for (int c = k; c < 20; ++c) pr[c] = sA[(p * 20 + c) * ST];
        if (p != k) {
#pragma unroll
            for (int c = k; c < 20; ++c) sA[(p * 20 + c) * ST] = A_(k, c);
        }
        double inv = 1.0 / pr[k];
#pragma unroll
        for (int c = k; c < 20; ++c) {
            pr[c] *= inv;
            A_(k, c) = pr[c];
        }
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == k || (r < k && r < 4)) continue;
            double f = A_(r, k);
#pragma unroll
            for (int c = k + 1; c < 20; ++c) A_(r, c) -= f * pr[c];
        }
    }
    // rows 4..9 <-> x^2 z, x^2, y^2 z, y^2, x y z, x y; columns 10..19 <->
    // x z^2, x z, x, y z^2, y z, y, z^3, z^2, z, 1.
    // z*row(m) - row(m z):  x*p3(z) + y*q3(z) + r4(z) = 0, stored descending.
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int k = 0; k < 13; ++k) bm[i][k] = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            bm[i][k] += A_(5 + 2 * i, 10 + k);
            bm[i][4 + k] += A_(5 + 2 * i, 13 + k);
            bm[i][1 + k] -= A_(4 + 2 * i, 10 + k);
            bm[i][5 + k] -= A_(4 + 2 * i, 13 + k);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            bm[i][8 + k] += A_(5 + 2 * i, 16 + k);
            bm[i][9 + k] -= A_(4 + 2 * i, 16 + k);
        }
    }
#undef A_
    // ---- determinant polynomial (degree 10, ascending coefficients)
#pragma unroll
    for (int i = 0; i < 11; ++i) cf[i] = 0;
    {
        // ascending views: column 0/1 cubic, column 2 quartic
        double f[3][3][5];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                f[r][0][k] = bm[r][3 - k];
                f[r][1][k] = bm[r][7 - k];
            }
            f[r][0][4] = f[r][1][4] = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) f[r][2][k] = bm[r][12 - k];
        }
        const int perm[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
        const double sgn[6] = {1, 1, 1, -1, -1, -1};
#pragma unroll
        for (int p = 0; p < 6; ++p) {
            double t01[9];
#pragma unroll
            for (int i = 0; i < 9; ++i) t01[i] = 0;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) t01[i + j] += f[0][perm[p][0]][i] * f[1][perm[p][1]][j];
#pragma unroll
            for (int i = 0; i < 9; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
                    if (i + j <= 10) cf[i + j] += sgn[p] * (t01[i] * f[2][perm[p][2]][j]);
        }
    }
    double cmax = 0;
#pragma unroll
    for (int i = 0; i <= 10; ++i) cmax = fmax(cmax, fabs(cf[i]));
    return cmax > 0 && cmax < DBL_MAX;
}

// Start values of one candidate: (x, y) from the null vector of the 3x3 system at the root
// z (best-conditioned row cross product).  Returns false when there is none.
__host__ __device__ __forceinline__ bool model_start(const double (*bm)[13], double z1, double* x, double* y) {
    double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
    double bz[3][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        bz[j][0] = bm[j][0] * z3 + bm[j][1] * z2 + bm[j][2] * z1 + bm[j][3];
        bz[j][1] = bm[j][4] * z3 + bm[j][5] * z2 + bm[j][6] * z1 + bm[j][7];
        bz[j][2] = bm[j][8] * z4 + bm[j][9] * z3 + bm[j][10] * z2 + bm[j][11] * z1 + bm[j][12];
    }
    double bx = 0, by = 0, bw = 0, bn = -1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int r0 = a == 2 ? 1 : 0, r1 = a == 0 ? 1 : 2;
        double cx = bz[r0][1] * bz[r1][2] - bz[r0][2] * bz[r1][1];
        double cy = bz[r0][2] * bz[r1][0] - bz[r0][0] * bz[r1][2];
        double cw = bz[r0][0] * bz[r1][1] - bz[r0][1] * bz[r1][0];
        double nn = cx * cx + cy * cy + cw * cw;
        if (nn > bn) { bn = nn; bx = cx; by = cy; bw = cw; }
    }
    if (!(bn > 0)) return false;
    double inv = 1.0 / sqrt(bn);
    bx *= inv; by *= inv; bw *= inv;
    if (fabs(bw) < 1e-10) return false;
    *x = bx / bw;
    *y = by / bw;
    return true;
}

// x1, x2: 5 normalised point pairs (x,y).  Writes up to 10 unit-norm essential matrices
// (row-major) to Es.  One lane does everything (host harness; the device path splits the
// root search and the refinement over sixteen lanes, see solve_back).
__host__ __device__ int five_point_dev(const double* x1, const double* x2, double* __restrict__ Es, double* sA,
                                       int chart = 0) {
    double EE[4][9], bm[3][13], cf[11];
    if (!five_point_front(x1, x2, sA, EE, bm, cf, chart)) return 0;
    // the elimination matrix is no longer needed: its LDS column becomes the root lists
    int nr;
    {
        double cmax = 0;
#pragma unroll
        for (int i = 0; i <= 10; ++i) cmax = fmax(cmax, fabs(cf[i]));
        if (fabs(cf[10]) > 1e-14 * cmax) {
            nr = real_roots10(cf, sA);
        } else {   // degree-deficient polynomial (rare): generic path
            double roots[10];
            nr = real_roots(cf, roots);
            for (int k = 0; k < nr; ++k) sA[k * ST] = roots[k];
        }
    }
    int count = 0;
    for (int ri = 0; ri < nr && count < kMaxModels; ++ri) {
        double z1 = sA[ri * ST], sx, sy;
        if (!model_start(bm, z1, &sx, &sy)) continue;
        double Ev[9];
        double res = polish_model_dev(EE, sx, sy, z1, Ev);
        if (!(res <= 1e-12)) continue;   // did not settle on a solution
        bool dup = false;               // two starts that settled on the same solution
        for (int m = 0; m < count && !dup; ++m) {
            double dp = 0, dm = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                dp = fmax(dp, fabs(Es[m * 9 + k] - Ev[k]));
                dm = fmax(dm, fabs(Es[m * 9 + k] + Ev[k]));
            }
            dup = fmin(dp, dm) < 1e-7;
        }
        if (dup) continue;
#pragma unroll
        for (int k = 0; k < 9; ++k) Es[count * 9 + k] = Ev[k];
        ++count;
    }
    return count;
}

// ---- root search and refinement, work items flattened over a workgroup's hypotheses
// After the elimination (one lane per hypothesis) what is left of a hypothesis is the root
// search - nine derivative levels, each a safeguarded Newton per monotone piece - and the
// refinement of up to ten candidates.  A polynomial of degree ten has few real roots, so a
// hypothesis offers only a handful of pieces per level and three or four candidates: the
// back end therefore takes HB hypotheses per workgroup, numbers all their pieces (then all
// their candidates) consecutively and hands one to each lane.  The outcome equals the
// one-lane routines' (same operations per piece, the same ordered bookkeeping).
constexpr int HB = 64;            // hypotheses per workgroup of the back end
constexpr int BACK_THREADS = 512;   // eight wavefronts: one sweep covers a level's pieces, two waves per SIMD overlap
// What the front end hands over: determinant polynomial, the three reduced constraint rows,
// the null-space basis; [0] = 1 if usable.
constexpr int HYP_REC = 88;   // doubles: flag, cf[11], bm[3][13], EE[4][9], one pad
constexpr int PIECES = 11;    // at most ten roots of the next derivative -> eleven pieces

template <int NHR>               // hypotheses per workgroup (at most 64: one wavefront numbers the pieces)
struct RootsSharedT {            // the nine derivative levels
    static constexpr int kHyp = NHR;
    double cf[NHR][11];
    double list[NHR][2][12];     // the two alternating root lists
    double px[NHR][PIECES];      // bracketed root of a piece
    double bound[NHR];           // root bound R of the current level
    uint8_t ptype[NHR][12];      // per piece: 1 left end is an exact zero, 2 bracketed root,
                                 // 4 right end is a touching extremum
    uint16_t map[NHR * PIECES];  // work item -> hypothesis | index << 8
    int16_t np[NHR];             // roots in the current list
    uint8_t deg[NHR];            // degree of the hypothesis' polynomial (0: nothing to solve)
    int total;
};
using RootsShared = RootsSharedT<HB>;
constexpr int HBM = 96;           // hypotheses per workgroup of the model stage on its own: 4.4 candidates each on
                                  // average, 420 of the 512 lanes busy (64 hypotheses leave 45 % of them idle, and at
                                  // 244 registers per lane nothing else is resident to use the slots)
// Gauss-Newton steps per refinement stage (they sum to kPolishSteps; {4, 8}, {2, 2, 8}, {2, 1, 1, 8},
// {3, 9} and {2, 10} measured within 3 % of this one, which was the fastest)
constexpr int kPolishStages = 3;
__device__ const int8_t kPolishPlan[kPolishStages] = {2, 1, 9};
template <int NH>
struct ModelsShared {            // roots -> refined, distinct essential matrices
    double Ev[NH][kMaxModels][9];
    uint8_t acc[NH][12];         // candidate settled on a solution
    uint16_t close[NH][12];      // bit m: candidate m < r settled within 1e-7 of candidate r
    uint16_t kept[NH];           // candidates that become models
    uint16_t map[NH * PIECES];   // work item -> hypothesis | index << 8
    int16_t np[NH];              // roots (candidates) of the hypothesis
    uint16_t strag[kPolishStages][NH * kMaxModels];   // candidates still moving after a stage (map entries)
    int nstrag[kPolishStages];
    int wsum;                    // candidates of the first 64 hypotheses
    int total;
};
struct BackShared {              // both stages in one workgroup (second-chart pass, debug entry)
    RootsShared r;
    ModelsShared<HB> m;
};

// Numbers the work items of the workgroup: hypothesis h offers cnt items.  Called by the
// first wavefront (lane = hypothesis); the map is valid after the next barrier.
template <class SH>
__device__ __forceinline__ void number_items(SH& sh, int h, int cnt) {
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int v = __shfl_up(incl, o);
        if (h >= o) incl += v;
    }
    const int off = incl - cnt;
    for (int g = 0; g < cnt; ++g) sh.map[off + g] = (uint16_t)(h | (g << 8));
    if (h == 63) sh.total = incl;
}

// The same for up to 128 hypotheses with sh.np[h] items each; called by the whole workgroup,
// the map is valid on return.
template <int NH, class SH>
__device__ __forceinline__ void number_items_wide(SH& sh) {
    static_assert(NH <= 128, "two wavefronts number the items");
    const int tid = threadIdx.x;
    int cnt = 0, incl = 0;
    if (tid < 128) {
        cnt = tid < NH ? sh.np[tid] : 0;
        incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int v = __shfl_up(incl, o);
            if ((tid & 63) >= o) incl += v;
        }
        if (tid == 63) sh.wsum = incl;
    }
    __syncthreads();
    if (tid < 128) {
        const int off = incl - cnt + (tid >= 64 ? sh.wsum : 0);
        for (int g = 0; g < cnt; ++g) sh.map[off + g] = (uint16_t)(tid | (g << 8));
        if (tid == 127) sh.total = off + cnt;
    }
    __syncthreads();
}

// One piece g of hypothesis h on the derivative level of degree DEG: list `si` holds the roots of the
// next derivative; records what the piece contributes (ptype) and its bracketed root (px).
template <int DEG, class RS>
__device__ __forceinline__ void root_piece(RS& sh, int si, int h, int g) {
    constexpr int J = 10 - DEG;
    const int np = sh.np[h];
    const int e = (int)sh.deg[h] - J;   // 1 .. DEG
    double d[DEG + 1], dd[DEG];
#pragma unroll
    for (int i = 0; i <= DEG; ++i) d[i] = sh.cf[h][i + J] * falling(i + J, J);
#pragma unroll
    for (int i = 0; i < DEG; ++i) dd[i] = d[i + 1] * (double)(i + 1);
    // Cauchy bound on the roots (one division; any bound serves, the roots do not depend on it)
    double R = 0, lead = d[DEG];
#pragma unroll
    for (int i = 0; i < DEG; ++i) {
        if (i < e) R = fmax(R, fabs(d[i]));
        if (i == e) lead = d[i];
    }
    R = R / fabs(lead) * (1.0 + 1e-12) + 1.0;
    if (g == 0) sh.bound[h] = R;
    const double* sP = sh.list[h][si];
    // Inner pieces lie between two roots of the derivative.  The two outer pieces are
    // unbounded: whether they hold a root follows from the sign at the inner end and the
    // sign at infinity, and the far end of the bracket is found by stepping outwards with
    // a step that grows fourfold (Newton from the middle of [end, +-R] would crawl, it
    // gains only a factor (DEG-1)/DEG per step while far from the roots).
    const bool left_open = g == 0, right_open = g == np;
    double a, b, fa, fb;
    if (!left_open) {
        a = fmin(fmax(sP[g - 1], -R), R);
        fa = horner<DEG>(d, a);
    }
    if (!right_open) {
        b = fmin(fmax(sP[g], -R), R);
        fb = horner<DEG>(d, b);
    }
    const double bT = right_open ? 0.0 : b, fbT = right_open ? 0.0 : fb;   // inner right end, for the extremum test
    const bool lead_neg = lead < 0;
    bool at_pivot = false;   // monotone polynomial with a zero exactly at the pivot
    if (left_open || right_open) {
        double p, fp, dir;
        if (left_open && right_open) {   // no extremum at all: monotone, pivot at 0
            p = 0.0;
            fp = d[0];
            dir = ((fp < 0) != lead_neg) ? 1.0 : -1.0;   // towards the sign opposite to fp
            at_pivot = fp == 0.0;
        } else if (right_open) {
            p = a; fp = fa; dir = 1.0;
        } else {
            p = b; fp = fb; dir = -1.0;
        }
        // sign of the polynomial at dir * infinity
        const bool inf_neg = dir > 0 ? lead_neg : (lead_neg != ((e & 1) != 0));
        double q = p, fq = fp;
        if (fp != 0.0 && ((fp < 0) != inf_neg)) {
            double step = 1e-3 * fmax(1.0, fabs(p));
            for (int it = 0; it < 64; ++it) {
                double x = p + dir * step;
                if (fabs(x) > R) x = dir * R;
                double fx = horner<DEG>(d, x);
                if (fx == 0.0 || ((fx < 0) != (fp < 0)) || fabs(x) >= R) {
                    p = q; fp = fq;   // last point on the near side
                    q = x; fq = fx;
                    break;
                }
                q = x; fq = fx;
                step *= 4.0;
            }
        }
        // [p, q] (in the direction dir) is the bracket, or p == q if there is no root
        if (dir > 0) { a = p; fa = fp; b = q; fb = fq; }
        else { a = q; fa = fq; b = p; fb = fp; }
    }
    const bool inc = b > a;
    const bool cA = inc && fa == 0.0 && !left_open;
    const bool cX = at_pivot || (inc && ((fa != 0.0 && ((fa < 0) != (fb < 0)) && fb != 0.0) ||
                                         (left_open && fa == 0.0) || (right_open && fb == 0.0)));
    if (cX) {
        double x;
        if (at_pivot) x = 0.0;
        else if (left_open && fa == 0.0) x = a;     // the probe hit the root exactly
        else if (right_open && fb == 0.0) x = b;
        else {
            double xl = fa < 0 ? a : b, xh = fa < 0 ? b : a;
            x = 0.5 * (a + b);
            double dxold = fabs(b - a), dx = dxold;
            double f = horner<DEG>(d, x), df = horner<DEG - 1>(dd, x);
            // The roots of the derivatives only delimit pieces and the final roots are
            // refined afterwards, so 1e-10 (1e-12 on the last level) relative is enough;
            // pieces that converge linearly (a root next to an extremum) stop at 40 steps.
            constexpr double tol = DEG < 10 ? 1e-10 : 1e-12;
            for (int it = 0; it < 40; ++it) {
                if ((((x - xh) * df - f) * ((x - xl) * df - f) > 0.0) || (fabs(2.0 * f) > fabs(dxold * df))) {
                    dxold = dx;
                    dx = 0.5 * (xh - xl);
                    x = xl + dx;
                } else {
                    dxold = dx;
                    dx = f / df;
                    x -= dx;
                }
                if (fabs(dx) <= tol * fmax(1.0, fabs(x))) break;
                f = horner<DEG>(d, x);
                df = horner<DEG - 1>(dd, x);
                if (f == 0.0) break;
                if (f < 0.0) xl = x; else xh = x;
            }
        }
        sh.px[h][g] = x;
    }
    bool cT = false;
    if (DEG == 10 && !right_open && fbT != 0.0) {
        // a local extremum within 1e-4 of zero, relative to the polynomial's magnitude there: a
        // double root or a pair of close roots lost to the elimination's rounding - a candidate
        double ab = fabs(bT), mag = fabs(d[DEG]);
#pragma unroll
        for (int i = DEG - 1; i >= 0; --i) mag = mag * ab + fabs(d[i]);
        cT = fabs(fbT) <= 1e-4 * mag;
    }
    sh.ptype[h][g] = (uint8_t)((cA ? 1 : 0) | (cX ? 2 : 0) | (cT ? 4 : 0));
}

// The ordered bookkeeping of the one-lane routine for hypothesis h: the pieces' contributions in
// ascending order become list 1 - si.
template <int DEG, class RS>
__device__ __forceinline__ void root_collect(RS& sh, int si, int h) {
const int np = sh.np[h];
    const double R = sh.bound[h];
    const double* sP = sh.list[h][si];
    double* sC = sh.list[h][1 - si];
    int nc = 0;
    double last = 0;
    for (int g = 0; g <= np; ++g) {
        const int ty = sh.ptype[h][g];
        if (ty == 0) continue;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (!((ty >> t) & 1)) continue;
            double v;
            if (t == 0) v = fmin(fmax(sP[g - 1], -R), R);    // never set for the first piece
            else if (t == 1) v = sh.px[h][g];
            else v = fmin(fmax(sP[g], -R), R);               // never set for the last piece
            const bool near = nc > 0 && fabs(v - last) <= 1e-7 * fmax(1.0, fabs(v));
            bool push;
            if (t == 0) push = nc == 0 || last != v;
            else if (t == 1) push = !(DEG == 10 && near);
            // a touching extremum is the double root itself only when neither neighbouring
            // piece holds a bracketed root (next to one, the bracketed roots are the two solutions)
            else push = !near && !(ty & 2) && !(sh.ptype[h][g + 1] & 2);
            if (push && nc < 10) {
                sC[nc] = v;
                last = v;
                ++nc;
            }
        }
    }
    sh.np[h] = (int16_t)nc;
}

// One derivative level (degree DEG) for the whole workgroup: list `si` holds the roots of the
// next derivative, the roots of this one go to list 1 - si (ascending).
template <int DEG, int T, class RS>   // T: threads of the workgroup
__device__ __forceinline__ void roots_level_block(RS& sh, int si) {
    const int tid = threadIdx.x;
    constexpr int J = 10 - DEG;
    constexpr int NHR = RS::kHyp;
    // A polynomial of degree n < 10 (its upper coefficients are stored as zeros) joins at the
    // level where its J-th derivative is linear - one unbounded piece, no list yet - and from
    // there on is a polynomial of degree e = n - J inside the arrays of degree DEG.
    if (tid < 64) number_items(sh, tid, tid < NHR && (int)sh.deg[tid] - J >= 1 ? sh.np[tid] + 1 : 0);
    __syncthreads();
    const int total = sh.total;
    for (int item = tid; item < total; item += T) {
        root_piece<DEG>(sh, si, sh.map[item] & 0xFF, sh.map[item] >> 8);
    }
    __syncthreads();
    if (tid < NHR && (int)sh.deg[tid] - J >= 1) root_collect<DEG>(sh, si, tid);
    __syncthreads();
}

// Front end of one hypothesis per lane (ST lanes per workgroup, the elimination matrix in
// LDS): writes the hypothesis record.
__device__ __forceinline__ void solve_front(bool active, const double* x1, const double* x2, double* sA,
                                            double* __restrict__ rec, int chart = 0) {
    bool ok = false;
    if (active) {
        double EE[4][9], bm[3][13], cf[11];
        ok = five_point_front(x1, x2, sA + threadIdx.x, EE, bm, cf, chart);
        if (ok) {
#pragma unroll
            for (int i = 0; i < 11; ++i) rec[1 + i] = cf[i];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int k = 0; k < 13; ++k) rec[12 + r * 13 + k] = bm[r][k];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 9; ++k) rec[51 + t * 9 + k] = EE[t][k];
        }
    }
    rec[0] = ok ? 1.0 : 0.0;
}

// Back end for up to HB hypotheses (records rec0 + h * HYP_REC, h < count), in two stages that
// are separate kernels in the RANSAC rounds: the root stage needs some 100 vector registers
// and little LDS, the model stage 244 registers (the null-space basis and the reduced
// constraints of a candidate stay in registers through the refinement) - fused, the root stage
// ran at the model stage's occupancy of one 512-lane workgroup per CU.
//
// Stage 1: real roots of the hypotheses' degree-10 polynomials; on return sh.list[h][1][0..np[h])
// holds them in ascending order.
// Takes hypothesis h's polynomial out of its record: coefficients, degree, the root of the ninth
// derivative as the first list.
template <class RS>
__device__ __forceinline__ void roots_init(RS& sh, int h, int count, const double* __restrict__ rec0) {
    {
        const double* rec = rec0 + (size_t)h * HYP_REC;
        int np = 0, n = 0;
        if (h < count && rec[0] != 0.0) {
            double cf[11], cmax = 0;
#pragma unroll
            for (int i = 0; i < 11; ++i) {
                cf[i] = rec[1 + i];
                cmax = fmax(cmax, fabs(cf[i]));
            }
            if (cmax > 0 && cmax < DBL_MAX) {
                // degree: upper coefficients that vanish against the largest one do not count (a
                // degenerate sample can make the leading one exactly zero).  Such a polynomial goes
                // through the same levels as the others: a single lane solving it the generic way
                // takes 200 us and holds its whole workgroup - and, in a round that fills the GPU
                // once, the round.
                n = 10;
#pragma unroll
                for (int i = 10; i >= 1; --i)
                    if (n == i && fabs(cf[i]) <= 1e-14 * cmax) n = i - 1;
#pragma unroll
                for (int i = 0; i < 11; ++i) sh.cf[h][i] = i <= n ? cf[i] : 0.0;
                if (n == 10) {
                    sh.list[h][0][0] = -(cf[9] * falling(9, 9)) / (cf[10] * falling(10, 9));   // root of the 9th derivative
                    np = 1;
                }
            }
        }
        sh.np[h] = (int16_t)np;
        sh.deg[h] = (uint8_t)n;
    }
}

template <int T, class RS>
__device__ __forceinline__ void roots_block(RS& sh, int count, const double* __restrict__ rec0) {
    if ((int)threadIdx.x < RS::kHyp) roots_init(sh, threadIdx.x, count, rec0);
    __syncthreads();
    roots_level_block<2, T>(sh, 0);
    roots_level_block<3, T>(sh, 1);
    roots_level_block<4, T>(sh, 0);
    roots_level_block<5, T>(sh, 1);
    roots_level_block<6, T>(sh, 0);
    roots_level_block<7, T>(sh, 1);
    roots_level_block<8, T>(sh, 0);
    roots_level_block<9, T>(sh, 1);
    roots_level_block<10, T>(sh, 0);   // the roots end up in list 1, their number in np
}

// The same stage one wavefront at a time: a workgroup is a single wavefront that takes RW_H
// hypotheses with RW_L lanes each; lane j of a hypothesis runs its pieces j, j + RW_L, ... (a
// polynomial of degree ten rarely offers more than eight pieces on a level), the first lane of a
// hypothesis does the ordered bookkeeping.  The barriers between the passes then are a
// wavefront's own (the hardware has nothing to wait for), wavefronts progress independently and
// six of them per SIMD hide one another's dependent chains - where 512 lanes behind workgroup
// barriers waited nine times for the slowest of some 220 pieces and for 64 lanes' bookkeeping.
constexpr int RW_H = 8, RW_L = 8;
template <int DEG, class RS>
__device__ __forceinline__ void roots_level_wave(RS& sh, int si) {
    constexpr int J = 10 - DEG;
    const int h = threadIdx.x / RW_L, sub = threadIdx.x % RW_L;
    const bool on = (int)sh.deg[h] - J >= 1;
    const int items = on ? sh.np[h] + 1 : 0;
    for (int g = sub; g < items; g += RW_L) root_piece<DEG>(sh, si, h, g);
    __syncthreads();
    if (sub == 0 && on) root_collect<DEG>(sh, si, h);
    __syncthreads();
}

__global__ __launch_bounds__(RW_H * RW_L, 6) void solve_roots_wave_kernel(const RState* __restrict__ st, int hr,
                                                                         const double* __restrict__ recs,
                                                                         double* __restrict__ roots,
                                                                         int32_t* __restrict__ nroots) {
    static_assert(RW_H * RW_L == 64, "one wavefront per workgroup");
    __shared__ RootsSharedT<RW_H> sh;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (st[b].done) return;
    const int h0 = blockIdx.x * RW_H, count = min(RW_H, hr - h0);
    const size_t slot = (size_t)b * hr + h0;
    if (tid < RW_H) roots_init(sh, tid, count, recs + slot * HYP_REC);
    __syncthreads();
    roots_level_wave<2>(sh, 0);
    roots_level_wave<3>(sh, 1);
    roots_level_wave<4>(sh, 0);
    roots_level_wave<5>(sh, 1);
    roots_level_wave<6>(sh, 0);
    roots_level_wave<7>(sh, 1);
    roots_level_wave<8>(sh, 0);
    roots_level_wave<9>(sh, 1);
    roots_level_wave<10>(sh, 0);
    for (int i = tid; i < count * kMaxModels; i += RW_H * RW_L) {
        const int h = i / kMaxModels, k = i - h * kMaxModels;
        if (k < sh.np[h]) roots[(slot + h) * kMaxModels + k] = sh.list[h][1][k];
    }
    if (tid < count) nroots[slot + tid] = sh.np[tid];
}

// Stage 2: the roots become models.  roots[h * roots_stride + r], r < sh.np[h] (set by the caller,
// visible after a barrier); writes the models of hypothesis h to Es0 + h * kMaxModels * 9 and
// their number to nm0[h].
// literal != 0 (study switch, VO_FIVE_POINT_LITERAL): the models are taken from the roots as they
// stand - no refinement, no residual gate, no duplicate drop - as OpenCV's solver leaves them.
template <int NH, int T>   // T: threads of the workgroup (at least 128, see number_items_wide)
__device__ __forceinline__ void models_block(ModelsShared<NH>& sh, int count, const double* __restrict__ rec0,
                                             const double* roots, int roots_stride, double* __restrict__ Es0,
                                             int32_t* __restrict__ nm0, int literal) {
    const int tid = threadIdx.x;
    // ---- one lane per candidate: start values, refinement, acceptance
    number_items_wide<NH>(sh);
    const int total = sh.total;
    // Refinement in stages: a wavefront runs as long as its slowest lane, most candidates have
    // converged after two Gauss-Newton steps (the start values come from roots good to ten digits),
    // some need one or two more, and one in a few dozen wanders until the budget of twelve is
    // spent (a spurious root).  After each stage the candidates still moving are collected (their
    // (x,y,z) parked in the model slot they will fill) and the next stage runs them on densely
    // packed lanes; polish_steps can be resumed, so the outcome is that of one uninterrupted run.
    const int8_t* const stage_steps = kPolishPlan;
    if (tid < kPolishStages) sh.nstrag[tid] = 0;
    __syncthreads();
    for (int item = tid; item < total; item += T) {
        const int h = sh.map[item] & 0xFF, r = sh.map[item] >> 8;
        const double* rec = rec0 + (size_t)h * HYP_REC;
        double z1 = roots[(size_t)h * roots_stride + r];
        double EE[4][9], bm[3][13], Ev[9], sx, sy;
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int k = 0; k < 13; ++k) bm[q][k] = rec[12 + q * 13 + k];
        bool acc = false;
        if (model_start(bm, z1, &sx, &sy)) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 9; ++k) EE[t][k] = rec[51 + t * 9 + k];
            if (literal || polish_steps(EE, sx, sy, z1, stage_steps[0])) {
                double res = polish_finish(EE, sx, sy, z1, Ev);
                acc = literal ? res < 1e299 : res <= 1e-12;   // settled on a solution
                if (acc) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) sh.Ev[h][r][k] = Ev[k];
                }
            } else {
                sh.Ev[h][r][0] = sx;
                sh.Ev[h][r][1] = sy;
                sh.Ev[h][r][2] = z1;
                sh.strag[0][atomicAdd(&sh.nstrag[0], 1)] = sh.map[item];
            }
        }
        sh.acc[h][r] = acc ? 1 : 0;
    }
    for (int st = 1; st < kPolishStages; ++st) {
        if (stage_steps[st] == 0) break;   // workgroup-uniform
        __syncthreads();
        const bool last = st + 1 == kPolishStages || stage_steps[st + 1] == 0;
        const int n_left = sh.nstrag[st - 1];
        for (int i = tid; i < n_left; i += T) {
            const uint16_t entry = sh.strag[st - 1][i];
            const int h = entry & 0xFF, r = entry >> 8;
            const double* rec = rec0 + (size_t)h * HYP_REC;
            double EE[4][9], Ev[9];
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int k = 0; k < 9; ++k) EE[t][k] = rec[51 + t * 9 + k];
            double sx = sh.Ev[h][r][0], sy = sh.Ev[h][r][1], z1 = sh.Ev[h][r][2];
            if (polish_steps(EE, sx, sy, z1, stage_steps[st]) || last) {
                // (a candidate that used the whole budget is still judged by its residual)
                if (polish_finish(EE, sx, sy, z1, Ev) <= 1e-12) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) sh.Ev[h][r][k] = Ev[k];
                    sh.acc[h][r] = 1;
                }
            } else {
                sh.Ev[h][r][0] = sx;
                sh.Ev[h][r][1] = sy;
                sh.Ev[h][r][2] = z1;
                sh.strag[st][atomicAdd(&sh.nstrag[st], 1)] = entry;
            }
        }
    }
    __syncthreads();
    // Two starts that settled on the same solution count once: in root order, a candidate
    // falls if a kept earlier one is within 1e-7 (either sign).  The comparisons run one
    // lane per candidate, the ordered decision is bit arithmetic by one lane per hypothesis.
    for (int item = tid; item < total; item += T) {
        const int h = sh.map[item] & 0xFF, r = sh.map[item] >> 8;
        uint32_t bits = 0;
        if (sh.acc[h][r] && !literal) {
            double Ev[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) Ev[k] = sh.Ev[h][r][k];
            for (int m = 0; m < r; ++m) {
                if (!sh.acc[h][m]) continue;
                double dp = 0, dm = 0;
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    double em = sh.Ev[h][m][k];
                    dp = fmax(dp, fabs(em - Ev[k]));
                    dm = fmax(dm, fabs(em + Ev[k]));
                }
                if (fmin(dp, dm) < 1e-7) bits |= 1u << m;
            }
        }
        sh.close[h][r] = (uint16_t)bits;
    }
    __syncthreads();
    if (tid < NH && tid < count) {
        const int h = tid, nr = sh.np[h];
        uint32_t kept = 0;
        for (int r = 0; r < nr; ++r)
            if (sh.acc[h][r] && !(sh.close[h][r] & kept)) kept |= 1u << r;
        sh.kept[h] = (uint16_t)kept;
        nm0[h] = __popc(kept);
    }
    __syncthreads();
    for (int item = tid; item < total; item += T) {
        const int h = sh.map[item] & 0xFF, r = sh.map[item] >> 8;
        const uint32_t kept = sh.kept[h];
        if (!((kept >> r) & 1u)) continue;
        double* o = Es0 + ((size_t)h * kMaxModels + __popc(kept & ((1u << r) - 1u))) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = sh.Ev[h][r][k];
    }
}

// Both stages in one workgroup.
__device__ __forceinline__ void solve_back_block(BackShared& sh, int count, const double* __restrict__ rec0,
                                                 double* __restrict__ Es0, int32_t* __restrict__ nm0, int literal = 0) {
    roots_block<BACK_THREADS>(sh.r, count, rec0);
    if (threadIdx.x < HB) sh.m.np[threadIdx.x] = sh.r.np[threadIdx.x];
    __syncthreads();
    models_block<HB, BACK_THREADS>(sh.m, count, rec0, &sh.r.list[0][1][0], 2 * 12, Es0, nm0, literal);
}

__global__ __launch_bounds__(ST) void solve_front_kernel(const RState* __restrict__ st, const double* __restrict__ m,
                                                         int n_max, const int32_t* __restrict__ samples,
                                                         int samples_stride, int h0, int hr, double* __restrict__ recs) {
    __shared__ double sA[200 * ST];
    const int b = blockIdx.y, h = blockIdx.x * ST + threadIdx.x;
    const RState& s = st[b];
    if (s.done || h >= hr) return;   // a finished pair reads nothing of this round any more
    double x1[10], x2[10];
    const bool active = (h0 + h) < s.niters;
    if (active) {
        const int32_t* idx = samples + (size_t)b * samples_stride + (size_t)h * 5;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double* p = m + ((size_t)b * n_max + idx[i]) * 4;
            x1[i * 2] = p[0]; x1[i * 2 + 1] = p[1];
            x2[i * 2] = p[2]; x2[i * 2 + 1] = p[3];
        }
    }
    solve_front(active, x1, x2, sA, recs + ((size_t)b * hr + h) * HYP_REC);
}

// Second pass over the hypotheses that came out of the first with an odd number of models.
// Real solutions of the five-point problem come in pairs (the complex roots of the degree-10
// polynomial are conjugate pairs), so an odd count says that one was lost: an ill-conditioned
// elimination turned a close real pair into a complex one, or threw a root too far for the
// refinement.  Those hypotheses (a few per thousand; the first back-end pass lists their slots
// b * hr + h in retry_list[1 ..], count in retry_list[0]) are solved again in a second chart of
// the null space, which conditions the same problem differently, and merge_models_kernel adds
// what that finds beyond the first pass.  Entry i of the list gets record i and model block i.
__global__ __launch_bounds__(ST) void retry_front_kernel(const double* __restrict__ m, int n_max,
                                                         const int32_t* __restrict__ samples, int samples_stride, int hr,
                                                         const int32_t* __restrict__ retry_list, double* __restrict__ recs) {
    __shared__ double sA[200 * ST];
    const int cnt = retry_list[0];
    for (int base = blockIdx.x * ST; base < cnt; base += gridDim.x * ST) {
        const int i = base + threadIdx.x;
        if (i >= cnt) continue;   // (no barrier below: every lane works in its own LDS column)
        const int slot = retry_list[1 + i], b = slot / hr, h = slot - b * hr;
        double x1[10], x2[10];
        const int32_t* idx = samples + (size_t)b * samples_stride + (size_t)h * 5;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double* p = m + ((size_t)b * n_max + idx[k]) * 4;
            x1[k * 2] = p[0]; x1[k * 2 + 1] = p[1];
            x2[k * 2] = p[2]; x2[k * 2 + 1] = p[3];
        }
        solve_front(true, x1, x2, sA, recs + (size_t)i * HYP_REC, 1);
    }
}

// Appends to a hypothesis' models those of the second pass that are not within 1e-7 (either
// sign) of one it already has.  One lane per second-pass entry: entry i belongs to slot
// retry_list[1 + i] (retry_list == nullptr: to slot i, `count` entries).
__global__ __launch_bounds__(256) void merge_models_kernel(const int32_t* __restrict__ retry_list, int count,
                                                           double* __restrict__ models, int32_t* __restrict__ nmodels,
                                                           const double* __restrict__ models2,
                                                           const int32_t* __restrict__ nmodels2) {
    const int cnt = retry_list ? retry_list[0] : count;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += gridDim.x * blockDim.x) {
        const size_t slot = retry_list ? (size_t)retry_list[1 + i] : (size_t)i;
        const int n2 = nmodels2[i];
        int n1 = nmodels[slot];
        if (n2 == 0 || !(n1 & 1)) continue;
        double* E1 = models + slot * kMaxModels * 9;
        const double* E2 = models2 + (size_t)i * kMaxModels * 9;
        for (int j = 0; j < n2 && n1 < kMaxModels; ++j) {
            bool dup = false;
            for (int k = 0; k < n1 && !dup; ++k) {
                double dp = 0, dm = 0;
                for (int q = 0; q < 9; ++q) {
                    dp = fmax(dp, fabs(E1[k * 9 + q] - E2[j * 9 + q]));
                    dm = fmax(dm, fabs(E1[k * 9 + q] + E2[j * 9 + q]));
                }
                dup = fmin(dp, dm) < 1e-7;
            }
            if (dup) continue;
            for (int q = 0; q < 9; ++q) E1[n1 * 9 + q] = E2[j * 9 + q];
            ++n1;
        }
        nmodels[slot] = n1;
    }
}


// (the root stage is a chain of short dependent passes between barriers, so what hides its latency
// is other workgroups: 80 registers, six wavefronts per SIMD)
template <int T, int NHR>
__global__ __launch_bounds__(T, 6) void solve_roots_kernel(const RState* __restrict__ st, int hr,
                                                                   const double* __restrict__ recs,
                                                                   double* __restrict__ roots,
                                                                   int32_t* __restrict__ nroots) {
    __shared__ RootsSharedT<NHR> sh;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (st[b].done) return;
    const int h0 = blockIdx.x * NHR, count = min(NHR, hr - h0);
    const size_t slot = (size_t)b * hr + h0;
    roots_block<T>(sh, count, recs + slot * HYP_REC);
    for (int i = tid; i < count * kMaxModels; i += T) {
        const int h = i / kMaxModels, k = i - h * kMaxModels;
        if (k < sh.np[h]) roots[(slot + h) * kMaxModels + k] = sh.list[h][1][k];
    }
    if (tid < count) nroots[slot + tid] = sh.np[tid];
}

template <int T, int NH>
__global__ __launch_bounds__(T) void solve_models_kernel(const RState* __restrict__ st, int hr,
                                                          const double* __restrict__ recs,
                                                          const double* __restrict__ roots,
                                                          const int32_t* __restrict__ nroots,
                                                          double* __restrict__ models,
                                                          int32_t* __restrict__ nmodels,
                                                          int32_t* __restrict__ retry_list, int literal) {
    __shared__ ModelsShared<NH> sh;
    const int b = blockIdx.y, tid = threadIdx.x;
    if (st[b].done) return;
    const int h0 = blockIdx.x * NH, count = min(NH, hr - h0);
    const size_t slot = (size_t)b * hr + h0;
    if (tid < NH) sh.np[tid] = tid < count ? (int16_t)nroots[slot + tid] : (int16_t)0;
    __syncthreads();
    models_block<NH, T>(sh, count, recs + slot * HYP_REC, roots + slot * kMaxModels, kMaxModels,
                        models + slot * kMaxModels * 9, nmodels + slot, literal);
    // odd model count: the hypothesis goes on the list of the second-chart pass (each lane reads
    // the count it wrote itself)
    if (retry_list && tid < count && (nmodels[slot + tid] & 1))
        retry_list[1 + atomicAdd(&retry_list[0], 1)] = (int32_t)(slot + tid);
}

__global__ __launch_bounds__(BACK_THREADS) void retry_back_kernel(const int32_t* __restrict__ retry_list,
                                                                  const double* __restrict__ recs,
                                                                  double* __restrict__ models2,
                                                                  int32_t* __restrict__ nmodels2) {
    __shared__ BackShared sh;
    const int cnt = retry_list[0];
    for (int base = blockIdx.x * HB; base < cnt; base += gridDim.x * HB) {   // workgroup-uniform
        solve_back_block(sh, min(HB, cnt - base), recs + (size_t)base * HYP_REC, models2 + (size_t)base * kMaxModels * 9,
                         nmodels2 + base);
        __syncthreads();
    }
}

#pragma clang fp contract(off)
// The inlier test (float)(err) <= t2 without the division in the common case: the
// double err = num / den rounds to a float <= t2 exactly when it lies below the midpoint
// t2_mid between t2 and the next float, so num against t2_mid * den decides unless the two are
// within 1e-15 relative (then the quotient is formed and rounded as written).
__device__ __forceinline__ bool sampson_inlier_dev(const double* E, double x1, double y1, double x2, double y2, float t2,
                                                   double t2_mid) {
    double Ex0 = E[0] * x1 + E[1] * y1 + E[2];
    double Ex1 = E[3] * x1 + E[4] * y1 + E[5];
    double Ex2 = E[6] * x1 + E[7] * y1 + E[8];
    double Et0 = E[0] * x2 + E[3] * y2 + E[6];
    double Et1 = E[1] * x2 + E[4] * y2 + E[7];
    double x2tEx1 = x2 * Ex0 + y2 * Ex1 + Ex2;
    double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    double num = x2tEx1 * x2tEx1, den = a + b + c + d;
    double r = t2_mid * den;
    if (num < r * (1.0 - 1e-15)) return true;
    if (num > r * (1.0 + 1e-15)) return false;
    return (float)(num / den) <= t2;
}

// One wavefront per hypothesis: every lane walks the points with stride 64, evaluates all of
// the hypothesis' models on each, and the per-lane inlier counts are summed across the wave.
//
// The decision must be the double-precision one (sampson_inlier_dev), but nearly every point
// is far from the threshold, and far from it single precision decides as well: the models go
// through packed single-precision fused multiply-adds two at a time (a fifth of the
// instructions of the double path, whose products and sums stay separate to match the CPU),
// with a bound on what rounding can have done to num - t2 * den; a test whose result is inside
// the bound goes on a per-wavefront list (one in a hundred on small-baseline pairs, where every
// near-skew model puts all points within a few pixels of their epipolar lines) and the lists are
// worked off in double precision by full wavefronts.
// The bound, with u = 2^-24, unit-norm E (|e_i| <= 1), M1 = 1 + |x1| + |y1|, M2 likewise:
// inputs and each fused step round by u, so the rows of E x1 are off by < 6 u M1, the epipolar
// product d by < 10 u M1 M2 (dd below takes 24 u M1 M2), num = d^2 by < 2 |d| dd + 3 dd^2 + u num,
// den by < 34 u max(M1, M2) sqrt(den) + 4 u den, which with t2 in front is below
// t2 (9 u M^2 + 36 u den); the sum is under  4 dd |d| + 64 u (num + t2 den) + 4 dd^2 + 16 u t2 (M1^2 + M2^2).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void score_kernel(const RState* __restrict__ st, const double* __restrict__ m,
                                                    const float* __restrict__ mf, int n_max, int hr,
                                                    const double* __restrict__ models,
                                                    const int32_t* __restrict__ nmodels, float t2, double t2_mid,
                                                    int32_t* __restrict__ counts, int h_lo, int h_hi) {
    constexpr int QCAP = 512;   // undecided tests kept per wavefront; an iteration adds at most 128
    __shared__ double sE[4][kMaxModels * 9];
    __shared__ __attribute__((aligned(8))) float sEf[4][kMaxModels / 2][9][2];   // element i of models 2p and 2p + 1
    __shared__ uint32_t s_q[4][QCAP];        // model << 24 | point
    __shared__ int s_extra[4][kMaxModels];   // inliers found among the listed tests
    int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (st[b].done) return;   // workgroup-uniform: the solve kernel wrote nothing for this pair
    int h = h_lo + blockIdx.x * 4 + wave;   // this launch scores the round's hypotheses h_lo .. h_hi - 1
    int nm = h < h_hi ? nmodels[(size_t)b * hr + h] : 0;   // wave-uniform
    const double* Eg = models + ((size_t)b * hr + h) * kMaxModels * 9;
    for (int i = lane; i < kMaxModels * 9; i += 64) {
        const double e = i < nm * 9 ? Eg[i] : 0.0;
        const int k = i / 9, q = i - k * 9;
        sE[wave][i] = e;
        sEf[wave][k >> 1][q][k & 1] = (float)e;
    }
    if (lane < kMaxModels) s_extra[wave][lane] = 0;
    __syncthreads();
    if (nm == 0) return;
    int n = st[b].n;
    // Only a hypothesis with more inliers than every earlier one changes anything, so one whose
    // models can no longer pass what the pair had when the round began (every point still to
    // come and every undecided test counted in its favour) is dropped: its counts stay partial,
    // below that mark.  The counts are wave-uniform (ballots), which makes the test scalar work.
    const int floor_cnt = max(st[b].max_good, 4);
    int cnt[kMaxModels];
#pragma unroll
    for (int k = 0; k < kMaxModels; ++k) cnt[k] = 0;
    int undecided = 0;   // tests ever put on the list
    bool dropped = false;
    const double* pb = m + (size_t)b * n_max * 4;
    const float* pf = mf + (size_t)b * n_max * 4;
    constexpr float U = 5.9604645e-8f;   // 2^-24
    const float t2m = (float)t2_mid;
    const f32x2 T2M = {t2m, t2m};
    uint32_t* q = s_q[wave];
    int qn = 0;   // wave-uniform
    auto flush = [&]() {
        for (int base = 0; base < qn; base += 64) {
            const int idx = base + lane;
            if (idx < qn) {
                const uint32_t ent = q[idx];
                const int k = (int)(ent >> 24);
                const double4 p = *reinterpret_cast<const double4*>(pb + (size_t)(ent & 0xFFFFFFu) * 4);
                if (sampson_inlier_dev(&sE[wave][k * 9], p.x, p.y, p.z, p.w, t2, t2_mid)) atomicAdd(&s_extra[wave][k], 1);
            }
        }
        qn = 0;
    };
    auto defer = [&](bool und, uint32_t ent) {   // appends the undecided lanes' entries in lane order
        const unsigned long long mk = __ballot(und);
        if (mk == 0) return;
        if (und) q[qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mk, 0u))] = ent;
        qn += (int)__popcll(mk);
        undecided += (int)__popcll(mk);
    };
    static_assert(kMaxModels <= 128 && QCAP >= 2 * 128, "list entry layout and room for one iteration");
    for (int i0 = 0; i0 < n; i0 += 64) {   // (all lanes stay in the loop: the list is a wavefront's)
        const int i = i0 + lane;
        const bool live = i < n;
        const float4 p = *reinterpret_cast<const float4*>(pf + (size_t)(live ? i : 0) * 4);
        const float x1 = p.x, y1 = p.y, x2 = p.z, y2 = p.w;
        const float m1 = 1.0f + fabsf(x1) + fabsf(y1), m2 = 1.0f + fabsf(x2) + fabsf(y2);
        const float dd = 24.0f * U * m1 * m2;
        const float c0 = 4.0f * dd * dd + 16.0f * U * t2m * (m1 * m1 + m2 * m2);
        const f32x2 X1 = {x1, x1}, Y1 = {y1, y1}, X2 = {x2, x2}, Y2 = {y2, y2};
        const f32x2 DD4 = {4.0f * dd, 4.0f * dd}, C0 = {c0, c0}, U64 = {64.0f * U, 64.0f * U};
#pragma unroll
        for (int pr = 0; pr < kMaxModels / 2; ++pr) {
            if (2 * pr >= nm) break;   // wave-uniform
            if (qn > QCAP - 128) flush();
            const f32x2* e = reinterpret_cast<const f32x2*>(&sEf[wave][pr][0][0]);
            const f32x2 a0 = __builtin_elementwise_fma(e[0], X1, __builtin_elementwise_fma(e[1], Y1, e[2]));
            const f32x2 a1 = __builtin_elementwise_fma(e[3], X1, __builtin_elementwise_fma(e[4], Y1, e[5]));
            const f32x2 a2 = __builtin_elementwise_fma(e[6], X1, __builtin_elementwise_fma(e[7], Y1, e[8]));
            const f32x2 b0 = __builtin_elementwise_fma(e[0], X2, __builtin_elementwise_fma(e[3], Y2, e[6]));
            const f32x2 b1 = __builtin_elementwise_fma(e[1], X2, __builtin_elementwise_fma(e[4], Y2, e[7]));
            const f32x2 d = __builtin_elementwise_fma(X2, a0, __builtin_elementwise_fma(Y2, a1, a2));
            const f32x2 num = d * d;
            const f32x2 den = __builtin_elementwise_fma(
                a0, a0, __builtin_elementwise_fma(a1, a1, __builtin_elementwise_fma(b0, b0, b1 * b1)));
            const f32x2 r = __builtin_elementwise_fma(-T2M, den, num);
            const f32x2 bound = __builtin_elementwise_fma(
                __builtin_elementwise_abs(d), DD4, __builtin_elementwise_fma(U64, __builtin_elementwise_fma(T2M, den, num), C0));
            // (written so that a NaN anywhere counts as undecided)
            const bool second = 2 * pr + 1 < nm;
            const bool und0 = live && !(fabsf(r.x) > bound.x), und1 = live && second && !(fabsf(r.y) > bound.y);
            cnt[2 * pr] += (int)__popcll(__ballot(live && !und0 && r.x < 0.0f));
            cnt[2 * pr + 1] += (int)__popcll(__ballot(live && second && !und1 && r.y < 0.0f));
            if (__any(und0 | und1)) {
                defer(und0, ((uint32_t)(2 * pr) << 24) | (uint32_t)i);
                defer(und1, ((uint32_t)(2 * pr + 1) << 24) | (uint32_t)i);
            }
        }
        int top = 0;
#pragma unroll
        for (int k = 0; k < kMaxModels; ++k) top = max(top, cnt[k]);
        if (top + undecided + max(n - i0 - 64, 0) <= floor_cnt) {   // wave-uniform
            dropped = true;
            break;
        }
    }
    if (!dropped) flush();
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < kMaxModels; ++k) {
            if (k >= nm) break;
            counts[((size_t)b * hr + h) * kMaxModels + k] = cnt[k] + s_extra[wave][k];
        }
    }
}

// A total order on essential matrices that does not depend on how they were found (OpenCV keeps
// the first of a hypothesis' models with the most inliers in the order its polynomial solver
// happens to emit them, which nothing else reproduces): E and -E are one model - the sign that
// makes the first entry of at least half the largest magnitude positive is compared -, entries
// are compared in storage order, differences below 1e-9 count as equal.
__device__ bool model_before(const double* A, const double* B) {
    double sa = 1.0, sb = 1.0, ma = 0.0, mb = 0.0;
    for (int i = 0; i < 9; ++i) {
        ma = fmax(ma, fabs(A[i]));
        mb = fmax(mb, fabs(B[i]));
    }
    for (int i = 8; i >= 0; --i) {
        if (fabs(A[i]) > 0.5 * ma) sa = A[i] < 0 ? -1.0 : 1.0;
        if (fabs(B[i]) > 0.5 * mb) sb = B[i] < 0 ? -1.0 : 1.0;
    }
    for (int i = 0; i < 9; ++i) {
        const double d = sa * A[i] - sb * B[i];
        if (fabs(d) > 1e-9) return d < 0;
    }
    return false;
}

// RANSAC bookkeeping over one round, then the next round's samples: one workgroup per pair.
// Both are serial by definition - the loop keeps the first model that beats every earlier one
// and shrinks its iteration bound as it goes; the samples come off one cv::RNG stream with
// rejection of repeated indices - and one lane replaying them took 190 us per 512-hypothesis
// round.  What is serial in fact is much less:
//  * a model changes the state only if it beats the best so far, and the bound is a monotone
//    function of the best count, so the replay reduces to a prefix maximum over the hypotheses'
//    best counts (one wavefront) and a serial walk over the few strict improvements;
//  * cv::RNG is a multiply-with-carry generator: with b = 2^32 and p = a b - 1 its 64-bit state
//    obeys s' = s / b (mod p), so the state k draws ahead is s b^-k mod p and b^-k mod p is what the
//    generator makes of the state 1 in k steps (d_mwc_jump, filled once per device).  The lanes
//    of one wavefront jump to the start of their 48-draw chunk and run the recurrence side by
//    side; all lanes reduce the draws modulo n and mark the stream positions whose next five
//    values are distinct; one lane then only walks set starts (a bit test per set, the rejection
//    loop where the bit is clear) and all lanes write the sets out.
constexpr int UPD_MAX_HR = 512;
constexpr int UPD_T = 256;
constexpr int UPD_CHUNK = 48, UPD_RAW = 64 * UPD_CHUNK;   // draws kept per round: 5 per set and room for repeats
constexpr unsigned long long kMwcA = 4164903690ull, kMwcP = (kMwcA << 32) - 1ull;
__device__ unsigned long long d_mwc_jump[64];   // the generator applied UPD_CHUNK * j times to the state 1

// x * y mod p by long division of the 128-bit product (once per lane and round)
__device__ unsigned long long mwc_mulmod(unsigned long long x, unsigned long long y) {
    const unsigned long long hi = __umul64hi(x, y), lo = x * y;
    unsigned long long rem = 0;
    for (int i = 127; i >= 0; --i) {
        const unsigned long long bit = i >= 64 ? (hi >> (i - 64)) & 1ull : (lo >> i) & 1ull;
        const bool carry = (rem >> 63) != 0;
        rem = (rem << 1) | bit;
        if (carry || rem >= kMwcP) rem -= kMwcP;
    }
    return rem;
}

// One 5-subset by the generator's rejection rule from reduced draws v[pos...]; returns the position
// after it, or -1 if the draws run out at `limit`.
__device__ __forceinline__ int take_set(const uint16_t* v, int pos, int limit, int* idx) {
    for (int i = 0; i < 5; ++i) {
        bool dup = true;
        int x = 0;
        while (dup) {
            if (pos >= limit) return -1;
            x = v[pos++];
            dup = false;
            for (int k = 0; k < i; ++k) dup |= idx[k] == x;
        }
        idx[i] = x;
    }
    return pos;
}

// The round's arrays hold arr_stride hypotheses per pair; this call replays the hr <= 512 of them
// from position arr_off on, which are the hypotheses h0 .. h0 + hr - 1 of the pair's sequence
// (a large fixed-count round is replayed in several calls; hr = 0 only draws).  The next_round
// sets drawn go to position out_off of the pair's sample block.  final_call != 0: no hypotheses follow.
__global__ __launch_bounds__(UPD_T) void update_kernel(RState* __restrict__ stv, int batch, RParams P, int h0, int hr,
                                                      int arr_stride, int arr_off,
                                                      const double* __restrict__ models, const int32_t* __restrict__ nmodels,
                                                      const int32_t* __restrict__ counts, int32_t* __restrict__ samples,
                                                      int samples_stride, int next_round, int out_off, int final_call,
                                                      int32_t* __restrict__ samples_log,
                                                      int log_cap, int32_t* __restrict__ retry_list) {
    __shared__ int32_t s_nm[UPD_MAX_HR];     // models of a hypothesis that take part
    __shared__ int32_t s_cnt[UPD_MAX_HR * kMaxModels];
    __shared__ int32_t s_hb[UPD_MAX_HR];     // a hypothesis' best count
    __shared__ uint8_t s_hk[UPD_MAX_HR];     // and the first model that reaches it
    __shared__ uint8_t s_imp[64];            // per lane of the first wavefront: which of its 8 hypotheses improve on all before
    __shared__ uint32_t s_raw[UPD_RAW];
    __shared__ uint16_t s_val[UPD_RAW];                 // draws modulo n (only used for n < 65536)
    __shared__ unsigned long long s_w5[UPD_RAW / 64];   // bit p: the five values from position p on are distinct
    __shared__ unsigned long long s_ck[65];             // generator state at the start of chunk j
    __shared__ uint16_t s_off[UPD_MAX_HR];              // stream position of a set's first draw
    __shared__ unsigned long long s_rng;
    __shared__ int s_drawn, s_stop, s_plain, s_sets, s_end;
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == 0 && tid == 0) retry_list[0] = 0;   // the round's second-chart list has been consumed
    if (stv[b].done) return;   // workgroup-uniform
    const int n = stv[b].n, niters0 = stv[b].niters, exact5 = stv[b].exact5;
    const int live = min(hr, max(niters0 - h0, 0));   // hypotheses the serial loop can reach
    for (int h = tid; h < UPD_MAX_HR; h += UPD_T) {
        int nm = h < live ? nmodels[(size_t)b * arr_stride + arr_off + h] : 0;
        if (exact5 && nm != 1) nm = 0;
        s_nm[h] = nm;
    }
    for (int i = tid; i < live * kMaxModels; i += UPD_T)
        s_cnt[i] = counts[((size_t)b * arr_stride + arr_off) * kMaxModels + i];
    if (samples_log) {
        const int32_t* src = samples + (size_t)b * samples_stride + (size_t)arr_off * 5;
        const int lim = min(live, log_cap - h0);
        for (int i = tid; i < lim * 5; i += UPD_T) samples_log[((size_t)b * log_cap + h0) * 5 + i] = src[i];
    }
    // The sample stream does not depend on the round's outcome - only how many sets are needed
    // does, and the iteration bound can only shrink during the replay - so the sets are drawn for
    // the bound as it stood before the replay; surplus sets are never used, and a pair that ends
    // in this round never looks at the stream again.
    const int want = next_round > 0 && !exact5 ? min(next_round, niters0 - (h0 + hr)) : 0;   // (five points: the one set is the points as they come)
    // (fewer than 16 points: most sets hold repeats; more than 65535: the values do not fit 16
    // bits - both rare in practice - then one lane draws the plain way)
    const bool fast_draw = want > 0 && n >= 16 && n < 65536;
    // draws to keep: five per set, a tenth more and a chunk for repeats; whole chunks
    const int R = min(UPD_RAW, (5 * want + (want >> 1) + 2 * UPD_CHUNK - 1) / UPD_CHUNK * UPD_CHUNK);
    if (tid == 0) {
        s_drawn = 0;
        s_plain = want > 0 && !fast_draw;
    }
    __syncthreads();
    for (int h = tid; h < UPD_MAX_HR; h += UPD_T) {
        int best = -1, bk = 0;
        const int nm = s_nm[h];
        for (int k = 0; k < nm; ++k) {
            const int good = s_cnt[h * kMaxModels + k];
            if (good > best) { best = good; bk = k; }
        }
        s_hb[h] = best;
        s_hk[h] = (uint8_t)bk;
    }
    __syncthreads();
    if (tid >= 64 && tid < 128 && fast_draw && (tid - 64) * UPD_CHUNK < R) {   // second wavefront: the raw stream, 48 draws per lane
        const int j = tid - 64;
        const unsigned long long g0 = stv[b].rng;
        unsigned long long g = j == 0 ? g0 : mwc_mulmod(g0, d_mwc_jump[j]);
        if (j > 0 && g < 2) s_plain = 1;   // the residue does not tell the states p, p + 1 from 0, 1 (never seen)
        s_ck[j] = g;
#pragma unroll 4
        for (int i = 0; i < UPD_CHUNK; ++i) s_raw[j * UPD_CHUNK + i] = rng_next(g);
        if ((j + 1) * UPD_CHUNK == R) s_ck[j + 1] = g;
    }
    // the replay, first wavefront: lane L owns hypotheses 8L .. 8L+7
    RState s;
    int best_h = -1;
    if (tid < 64) {
        int v[8], lane_max = -1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[j] = s_hb[8 * tid + j];
            lane_max = max(lane_max, v[j]);
        }
        int incl = lane_max;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int u = __shfl_up(incl, o);
            if (tid >= o) incl = max(incl, u);
        }
        int before = __shfl_up(incl, 1);
        s = stv[b];
        int cur = max(max(s.max_good, 4), tid > 0 ? before : -1);
        uint32_t bits = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (v[j] > cur) {
                cur = v[j];
                bits |= 1u << j;
            }
        s_imp[tid] = (uint8_t)bits;
    }
    __syncthreads();
    if (tid == 0) {
        // the strict improvements in order; between two of them the bound is constant
        int last = -1;
        bool ended = false;
        for (int L = 0; L < 64 && !ended; ++L) {
            uint32_t bits = s_imp[L];
            while (bits) {
                const int j = __ffs(bits) - 1;
                bits &= bits - 1;
                const int h = 8 * L + j;
                if (h0 + h >= s.niters) { ended = true; break; }   // the loop ended before it got here
                const int good = s_hb[h];
                s.max_good = good;
                best_h = h;
                last = h;
                if (P.fixed_iters <= 0 && !s.exact5)
                    s.niters = update_num_iters(P.confidence, (double)(s.n - good) / s.n, s.niters);
            }
        }
        // the hypothesis of an improvement is completed before the bound is looked at again
        const int stop = min(hr, max(max(s.niters - h0, 0), last + 1));
        if (stop > 0) s.iters_run = h0 + stop;
        if (best_h >= 0) {
            // among the winning hypothesis' models with equally many inliers: the first in an order
            // that does not depend on how the models were found (see model_before)
            const double* Eh = models + ((size_t)b * arr_stride + arr_off + best_h) * kMaxModels * 9;
            int k0 = s_hk[best_h];
            for (int k = k0 + 1; k < s_nm[best_h]; ++k)
                if (s_cnt[best_h * kMaxModels + k] == s_hb[best_h] && model_before(Eh + k * 9, Eh + k0 * 9)) k0 = k;
            for (int i = 0; i < 9; ++i) s.bestE[i] = Eh[k0 * 9 + i];
        }
        if (hr > 0 && (stop < hr || h0 + hr >= s.niters || final_call)) s.done = 1;   // (hr = 0: a call that only draws)
        s_stop = stop;
    }
    __syncthreads();
    if (tid < 64) {   // candidates of the hypotheses the loop walked
        const int stop = s_stop;
        int sum = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (8 * tid + j < stop) sum += s_nm[8 * tid + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (tid == 0) s.models_scored += sum;
    }
    int32_t* out = samples + (size_t)b * samples_stride + (size_t)out_off * 5;
    const bool plain = s_plain != 0;
    if (want > 0 && plain) {
        if (tid == 64) {
            RState d;
            d.rng = stv[b].rng;
            d.n = n;
            draw_samples(d, out, want);
            s_rng = d.rng;
            s_drawn = 1;
        }
    } else if (want > 0) {   // workgroup-uniform
        for (int i = tid; i < R; i += UPD_T) s_val[i] = (uint16_t)(s_raw[i] % (uint32_t)n);
        __syncthreads();
        for (int i = tid; i < ((R + 63) & ~63); i += UPD_T) {
            bool distinct = false;
            if (i + 5 <= R) {
                const int a0 = s_val[i], a1 = s_val[i + 1], a2 = s_val[i + 2], a3 = s_val[i + 3], a4 = s_val[i + 4];
                distinct = a0 != a1 && a0 != a2 && a0 != a3 && a0 != a4 && a1 != a2 && a1 != a3 && a1 != a4 && a2 != a3 &&
                           a2 != a4 && a3 != a4;
            }
            const unsigned long long m = __ballot(distinct);
            if ((tid & 63) == 0) s_w5[i >> 6] = m;
        }
        __syncthreads();
        if (tid >= 64 && tid < 128) {
            // Set starts, by the second wavefront.  Nearly every set is five distinct draws in a row,
            // so the starts run on in steps of five: lane j tests the set that starts 5 j draws on, the
            // leading run of regular ones is taken in one go, and only a set with a repeat (one in
            // n / 10) is walked by the rejection loop - some ten passes for 512 sets where one lane
            // stepped through them one by one (25 us of the kernel's 70).
            const int lane = tid - 64;
            int pos = 0, h = 0;   // wavefront-uniform
            while (h < want && pos < R) {
                const int p = pos + 5 * lane;
                const bool regular = h + lane < want && p < R && ((s_w5[p >> 6] >> (p & 63)) & 1ull);
                const unsigned long long m = __ballot(regular);
                const int run = m == ~0ull ? 64 : __ffsll((long long)~m) - 1;
                if (lane < run) s_off[h + lane] = (uint16_t)p;
                h += run;
                pos += 5 * run;
                if (run < 64 && h < want && pos < R) {
                    int idx[5];
                    const int nxt = take_set(s_val, pos, R, idx);   // (every lane walks the same draws)
                    if (nxt < 0) break;   // the kept draws run out inside this set
                    if (lane == 0) s_off[h] = (uint16_t)pos;
                    ++h;
                    pos = nxt;
                }
            }
            if (lane == 0) {
                s_sets = h;
                s_end = pos;
            }
        }
        __syncthreads();
        const int sets = s_sets;
        for (int h = tid; h < sets; h += UPD_T) {
            const int pos = s_off[h];
            int idx[5];
            if ((s_w5[pos >> 6] >> (pos & 63)) & 1ull) {
#pragma unroll
                for (int i = 0; i < 5; ++i) idx[i] = s_val[pos + i];
            } else {
                take_set(s_val, pos, R, idx);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) out[h * 5 + i] = idx[i];
        }
        if (tid == 64) {
            const int at = s_end;   // draws consumed by the complete sets
            unsigned long long g = s_ck[at / UPD_CHUNK];
            for (int i = 0; i < at % UPD_CHUNK; ++i) rng_next(g);
            if (sets < want) {   // more repeats than the kept draws cover: the rest the plain way
                RState d;
                d.rng = g;
                d.n = n;
                draw_samples(d, out + sets * 5, want - sets);
                g = d.rng;
            }
            s_rng = g;
            s_drawn = 1;
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (s_drawn) s.rng = s_rng;
        stv[b] = s;
    }
}

// ---- the bookkeeping of a fixed-count round
// With a fixed hypothesis count the iteration bound never moves, so the serial loop's outcome is
// simply the first hypothesis that reaches the round's largest inlier count (if that beats what
// earlier rounds found and the minimum of five): every 512-hypothesis piece of the round finds
// its own first maximum side by side (fixed_scan_kernel, a workgroup per piece and pair) and one
// lane per pair walks the pieces in order (fixed_pick_kernel) - two launches where replaying the
// pieces one after the other through update_kernel took eight for 4096 hypotheses.
struct FixedPart {
    int best;     // largest count of a model in the piece (-1: none)
    int h;        // first hypothesis of the round that reaches it
    int nm_sum;   // models of the piece's live hypotheses
    int pad;
};
constexpr int FX_T = 256;
__global__ __launch_bounds__(FX_T) void fixed_scan_kernel(const RState* __restrict__ stv, int h0, int hr,
                                                         const int32_t* __restrict__ nmodels,
                                                         const int32_t* __restrict__ counts,
                                                         FixedPart* __restrict__ parts,
                                                         const int32_t* __restrict__ samples, int samples_stride,
                                                         int32_t* __restrict__ samples_log, int log_cap, int piece0,
                                                         int n_pieces) {
    __shared__ unsigned long long s_key[FX_T / 64];
    __shared__ int s_sum[FX_T / 64];
    const int b = blockIdx.y, piece = piece0 + blockIdx.x, tid = threadIdx.x;
    if (stv[b].done) return;   // workgroup-uniform
    const int niters = stv[b].niters, exact5 = stv[b].exact5;
    const int base = piece * UPD_MAX_HR;
    const int live = min(min(UPD_MAX_HR, hr - base), max(niters - (h0 + base), 0));
    if (samples_log) {   // the sets the loop walks, in the order drawn
        const int32_t* src = samples + (size_t)b * samples_stride + (size_t)base * 5;
        const int lim = min(live, log_cap - (h0 + base));
        for (int i = tid; i < lim * 5; i += FX_T) samples_log[((size_t)b * log_cap + h0 + base) * 5 + i] = src[i];
    }
    // key: count + 1 in the upper half, the complement of the hypothesis index below it, so that the
    // largest key is the largest count at the lowest index
    unsigned long long key = 0;
    int sum = 0;
    for (int h = tid; h < live; h += FX_T) {
        const size_t slot = (size_t)b * hr + base + h;
        int nm = nmodels[slot];
        if (exact5 && nm != 1) nm = 0;
        sum += nm;
        int best = -1;
        for (int k = 0; k < nm; ++k) best = max(best, counts[slot * kMaxModels + k]);
        const unsigned long long mine = ((unsigned long long)(uint32_t)(best + 1) << 32) | (uint32_t)(0x7FFFFFFF - (base + h));
        key = mine > key ? mine : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
        sum += __shfl_xor(sum, o);
    }
    if ((tid & 63) == 0) {
        s_key[tid >> 6] = key;
        s_sum[tid >> 6] = sum;
    }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < FX_T / 64; ++w) {
            key = s_key[w] > key ? s_key[w] : key;
            sum += s_sum[w];
        }
        FixedPart p;
        p.best = (int)(uint32_t)(key >> 32) - 1;
        p.h = 0x7FFFFFFF - (int)(uint32_t)key;
        p.nm_sum = sum;
        p.pad = 0;
        parts[(size_t)b * n_pieces + piece] = p;
    }
}

// Pieces p_lo .. p_hi - 1 of the round's n_pieces; the call with p_hi == n_pieces closes the round.
__global__ void fixed_pick_kernel(RState* __restrict__ stv, int batch, int h0, int hr, int n_pieces, int p_lo, int p_hi,
                                  const FixedPart* __restrict__ parts, const double* __restrict__ models,
                                  const int32_t* __restrict__ nmodels, const int32_t* __restrict__ counts,
                                  int final_call, int32_t* __restrict__ retry_list) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    const bool closes = p_hi == n_pieces;
    if (b == 0 && closes) retry_list[0] = 0;   // the round's second-chart list has been consumed
    if (b >= batch) return;
    RState s = stv[b];
    if (s.done) return;
    const int stop = min(hr, max(s.niters - h0, 0));   // hypotheses of the round the serial loop walks
    int cur = max(s.max_good, 4), win = -1;
    for (int p = p_lo; p < p_hi; ++p) {
        const FixedPart q = parts[(size_t)b * n_pieces + p];
        s.models_scored += q.nm_sum;
        if (q.best > cur) {
            cur = q.best;
            win = q.h;
        }
    }
    if (win >= 0) {
        // among the winning hypothesis' models with equally many inliers: the first in an order that
        // does not depend on how the models were found (see model_before)
        const size_t slot = (size_t)b * hr + win;
        const double* Eh = models + slot * kMaxModels * 9;
        int nm = nmodels[slot];
        if (s.exact5 && nm != 1) nm = 0;
        int k0 = -1;
        for (int k = 0; k < nm; ++k)
            if (counts[slot * kMaxModels + k] == cur && (k0 < 0 || model_before(Eh + k * 9, Eh + k0 * 9))) k0 = k;
        s.max_good = cur;
        for (int i = 0; i < 9; ++i) s.bestE[i] = Eh[k0 * 9 + i];
    }
    if (closes) {
        if (stop > 0) s.iters_run = h0 + stop;
        if (stop < hr || h0 + hr >= s.niters || final_call) s.done = 1;
    }
    stv[b] = s;
}

// ------------------------------------------------------------------ pose recovery
// Smallest-eigenvalue eigenvector of a symmetric 4x4 (cyclic Jacobi).
__device__ void smallest_eigvec4(double S[4][4], double* v) {
    double V[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 12; ++sweep) {
        double offd = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = i + 1; j < 4; ++j) offd += S[i][j] * S[i][j];
        double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2] + S[3][3] * S[3][3];
        if (offd <= 1e-60 * diag || offd == 0) break;
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int q = p + 1; q < 4; ++q) {
                double apq = S[p][q];
                if (apq == 0.0) continue;
                double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double skp = S[k][p], skq = S[k][q];
                    S[k][p] = c * skp - s * skq;
                    S[k][q] = s * skp + c * skq;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = c * spk - s * sqk;
                    S[q][k] = s * spk + c * sqk;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int best = 0;
    double bv = S[0][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (S[i][i] < bv) { bv = S[i][i]; best = i; }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = best == 0 ? V[k][0] : best == 1 ? V[k][1] : best == 2 ? V[k][2] : V[k][3];
}

// The same eigenvector without the Jacobi sweeps when the smallest eigenvalue is well
// separated (every triangulation with usable parallax): the adjugate of S has the
// eigenvectors of S with eigenvalues prod_{j != i} lambda_j, so the wanted one is its
// dominant eigenvector and a few products adj(S) v converge at the rate lambda_1/lambda_2.
// Returns false (v untouched) when eight products do not settle to 4e-16; the caller then
// takes the Jacobi route.
__device__ __forceinline__ bool smallest_eigvec4_adj(const double S[4][4], double* v) {
    double C[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = i; j < 4; ++j) {
            // cofactor (i, j): rows without i, columns without j
            const int r0 = i == 0 ? 1 : 0, r1 = i <= 1 ? 2 : 1, r2 = i <= 2 ? 3 : 2;
            const int c0 = j == 0 ? 1 : 0, c1 = j <= 1 ? 2 : 1, c2 = j <= 2 ? 3 : 2;
            double d = S[r0][c0] * (S[r1][c1] * S[r2][c2] - S[r1][c2] * S[r2][c1]) -
                       S[r0][c1] * (S[r1][c0] * S[r2][c2] - S[r1][c2] * S[r2][c0]) +
                       S[r0][c2] * (S[r1][c0] * S[r2][c1] - S[r1][c1] * S[r2][c0]);
            C[i][j] = C[j][i] = ((i + j) & 1) ? -d : d;
        }
    // start from the column with the largest diagonal entry
    int best = 0;
    double bv = C[0][0];
#pragma unroll
    for (int i = 1; i < 4; ++i)
        if (C[i][i] > bv) { bv = C[i][i]; best = i; }
    if (!(bv > 0)) return false;
    double x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = best == 0 ? C[k][0] : best == 1 ? C[k][1] : best == 2 ? C[k][2] : C[k][3];
    double nrm = 1.0 / sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3]);
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] *= nrm;
    for (int it = 0; it < 8; ++it) {
        double y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = C[k][0] * x[0] + C[k][1] * x[1] + C[k][2] * x[2] + C[k][3] * x[3];
        double n2 = y[0] * y[0] + y[1] * y[1] + y[2] * y[2] + y[3] * y[3];
        if (!(n2 > 0)) return false;
        nrm = 1.0 / sqrt(n2);
        double dmax = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            y[k] *= nrm;
            dmax = fmax(dmax, fabs(y[k] - x[k]));
            x[k] = y[k];
        }
        if (dmax <= 4e-16) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = x[k];
            return true;
        }
    }
    return false;
}

// decomposeEssentialMat: E = U diag(s,s,0) V^T -> R1 = U W V^T, R2 = U W^T V^T, t = u3.
__device__ void decompose_essential_dev(const double* E, double* R1, double* R2, double* t) {
    // eigen-decomposition of E^T E (3x3) by Jacobi -> V
    double S[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            S[i][j] = E[0 * 3 + i] * E[0 * 3 + j] + E[1 * 3 + i] * E[1 * 3 + j] + E[2 * 3 + i] * E[2 * 3 + j];
            V[i][j] = i == j ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 16; ++sweep) {
        double offd = S[0][1] * S[0][1] + S[0][2] * S[0][2] + S[1][2] * S[1][2];
        double diag = S[0][0] * S[0][0] + S[1][1] * S[1][1] + S[2][2] * S[2][2];
        if (offd <= 1e-40 * diag) break;   // off-diagonal part below 1e-20 relative: converged in double
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double apq = S[p][q];
                if (apq == 0.0) continue;
                double theta = (S[q][q] - S[p][p]) / (2.0 * apq);
                double tt = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double skp = S[k][p], skq = S[k][q];
                    S[k][p] = c * skp - s * skq;
                    S[k][q] = s * skp + c * skq;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double spk = S[p][k], sqk = S[q][k];
                    S[p][k] = c * spk - s * sqk;
                    S[q][k] = s * spk + c * sqk;
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    // order eigenvalues descending
    int o0 = 0, o1 = 1, o2 = 2;
    double e0 = S[0][0], e1 = S[1][1], e2 = S[2][2];
    if (e1 > e0) { double tt = e0; e0 = e1; e1 = tt; int ti = o0; o0 = o1; o1 = ti; }
    if (e2 > e0) { double tt = e0; e0 = e2; e2 = tt; int ti = o0; o0 = o2; o2 = ti; }
    if (e2 > e1) { double tt = e1; e1 = e2; e2 = tt; int ti = o1; o1 = o2; o2 = ti; }
    double v1[3], v2[3], v3[3], u1[3], u2[3], u3[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { v1[k] = V[k][o0]; v2[k] = V[k][o1]; }
    v3[0] = v1[1] * v2[2] - v1[2] * v2[1];
    v3[1] = v1[2] * v2[0] - v1[0] * v2[2];
    v3[2] = v1[0] * v2[1] - v1[1] * v2[0];
    double n1 = 0, n2 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        u1[i] = E[i * 3] * v1[0] + E[i * 3 + 1] * v1[1] + E[i * 3 + 2] * v1[2];
        u2[i] = E[i * 3] * v2[0] + E[i * 3 + 1] * v2[1] + E[i * 3 + 2] * v2[2];
        n1 += u1[i] * u1[i];
        n2 += u2[i] * u2[i];
    }
    n1 = 1.0 / sqrt(n1);
#pragma unroll
    for (int i = 0; i < 3; ++i) u1[i] *= n1;
    // re-orthogonalise u2 against u1 (repeated singular value)
    double d12 = (u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2]);
    n2 = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { u2[i] -= d12 * u1[i]; n2 += u2[i] * u2[i]; }
    n2 = 1.0 / sqrt(n2);
#pragma unroll
    for (int i = 0; i < 3; ++i) u2[i] *= n2;
    u3[0] = u1[1] * u2[2] - u1[2] * u2[1];
    u3[1] = u1[2] * u2[0] - u1[0] * u2[2];
    u3[2] = u1[0] * u2[1] - u1[1] * u2[0];
    // U = [u1 u2 u3], V = [v1 v2 v3] both with det +1.
    // U W V^T  = u1 (-v2)^T... expand: W = [0 1 0; -1 0 0; 0 0 1]
    // (U W)   = [-u2, u1, u3];  (U W^T) = [u2, -u1, u3]
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            R1[i * 3 + j] = -u2[i] * v1[j] + u1[i] * v2[j] + u3[i] * v3[j];
            R2[i * 3 + j] = u2[i] * v1[j] - u1[i] * v2[j] + u3[i] * v3[j];
        }
    t[0] = u3[0]; t[1] = u3[1]; t[2] = u3[2];
}

// One workgroup of 1024 lanes per pair: inlier mask of the winning E, four-fold
// triangulation/cheirality test (one quarter of the workgroup per candidate pose), pose
// selection and validity rules.
constexpr int FIN_T = 256;   // lanes per candidate pose
__global__ __launch_bounds__(4 * FIN_T) void finish_kernel(const RState* __restrict__ stv, const double* __restrict__ m,
                                                           int n_max, RParams P, vo_pair_result* __restrict__ res,
                                                           double* __restrict__ E_out, uint8_t* __restrict__ mask_e,
                                                           uint8_t* __restrict__ mask_pose, int32_t* __restrict__ good_out) {
    __shared__ double sR[2][9], sT[3], sE[9];
    __shared__ int cnt[4], sel;
    const int b = blockIdx.x, tid = threadIdx.x, c = tid / FIN_T, j = tid % FIN_T;
    const RState& s = stv[b];
    const int n = s.n;
    vo_pair_result* r = res ? res + b : nullptr;
    if (tid < 4) cnt[tid] = 0;
    if (tid < 9) sE[tid] = s.bestE[tid];
    if (tid == 0) {
        if (good_out) good_out[b * 2] = s.max_good;
        if (good_out) good_out[b * 2 + 1] = s.iters_run;
        if (E_out) for (int k = 0; k < 9; ++k) E_out[b * 9 + k] = s.max_good > 0 ? s.bestE[k] : 0.0;
        if (r) {
            r->inliers = 0;
            r->valid = 0;
            r->ransac_iters = s.iters_run;
            r->models_scored = s.models_scored;
            for (int k = 0; k < 9; ++k) r->R[k] = 0;
            for (int k = 0; k < 3; ++k) r->t[k] = 0;
        }
    }
    if (s.max_good <= 0) {   // block-uniform
        if (mask_e) for (int i = tid; i < n; i += 4 * FIN_T) mask_e[(size_t)b * n_max + i] = 0;
        if (mask_pose) for (int i = tid; i < n; i += 4 * FIN_T) mask_pose[(size_t)b * n_max + i] = 0;
        return;
    }
    if (tid == 0) decompose_essential_dev(s.bestE, sR[0], sR[1], sT);
    __syncthreads();
    const double* pb = m + (size_t)b * n_max * 4;
    auto load_pose = [&](int cand, double (*Pm)[4]) {   // [R | +-t]
        const double* R = sR[cand & 1];
        const double sg = cand < 2 ? 1.0 : -1.0;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            Pm[a][0] = R[a * 3]; Pm[a][1] = R[a * 3 + 1]; Pm[a][2] = R[a * 3 + 2];
            Pm[a][3] = sg * sT[a];
        }
    };
    // triangulates point i against [I|0] and Pm; true if it is an inlier of E in front of both
    // cameras and nearer than 50 baselines
    auto in_front = [&](int i, const double (*Pm)[4], bool* inl_out) -> bool {
        double x1 = pb[i * 4], y1 = pb[i * 4 + 1], x2 = pb[i * 4 + 2], y2 = pb[i * 4 + 3];
        bool inl = s.exact5 ? true : sampson_inlier_dev(sE, x1, y1, x2, y2, P.t2, P.t2_mid);
        *inl_out = inl;
        if (!inl) return false;   // only inliers of E take part in the cheirality vote
        // DLT rows: x*P[2]-P[0], y*P[2]-P[1] for P0 = [I|0] and P
        double A[4][4];
        A[0][0] = -1; A[0][1] = 0; A[0][2] = x1; A[0][3] = 0;
        A[1][0] = 0; A[1][1] = -1; A[1][2] = y1; A[1][3] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            A[2][q] = x2 * Pm[2][q] - Pm[0][q];
            A[3][q] = y2 * Pm[2][q] - Pm[1][q];
        }
        double S[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                S[a][q] = A[0][a] * A[0][q] + A[1][a] * A[1][q] + A[2][a] * A[2][q] + A[3][a] * A[3][q];
        double X[4];
        if (!smallest_eigvec4_adj(S, X)) smallest_eigvec4(S, X);
        bool ok = X[2] * X[3] > 0;
        double Xn0 = X[0] / X[3], Xn1 = X[1] / X[3], Xn2 = X[2] / X[3];
        ok = ok && (Xn2 < 50.0);
        double z2 = Pm[2][0] * Xn0 + Pm[2][1] * Xn1 + Pm[2][2] * Xn2 + Pm[2][3];
        return ok && (z2 > 0) && (z2 < 50.0) && inl;
    };
    double Pm[3][4];
    load_pose(c, Pm);
    int mine = 0;
    for (int i = j; i < n; i += FIN_T) {
        bool inl;
        mine += in_front(i, Pm, &inl);
        if (mask_e && c == 0) mask_e[(size_t)b * n_max + i] = inl;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((tid & 63) == 0) atomicAdd(&cnt[c], mine);
    __syncthreads();
    if (tid == 0) {
        int g1 = cnt[0], g2 = cnt[1], g3 = cnt[2], g4 = cnt[3], sl;
        if (g1 >= g2 && g1 >= g3 && g1 >= g4) sl = 0;
        else if (g2 >= g1 && g2 >= g3 && g2 >= g4) sl = 1;
        else if (g3 >= g1 && g3 >= g2 && g3 >= g4) sl = 2;
        else sl = 3;
        sel = sl;
        int good = cnt[sl];
        if (r && good >= 10) {
            r->valid = 1;
            r->inliers = good;
            for (int k = 0; k < 9; ++k) r->R[k] = sR[sl & 1][k];
            for (int k = 0; k < 3; ++k) r->t[k] = (sl < 2 ? 1.0 : -1.0) * sT[k];
        }
    }
    if (!mask_pose) return;
    __syncthreads();
    // the winning pose's per-point outcome, recomputed by the whole workgroup
    load_pose(sel, Pm);
    for (int i = tid; i < n; i += 4 * FIN_T) {
        bool inl;
        mask_pose[(size_t)b * n_max + i] = in_front(i, Pm, &inl);
    }
}

// Five-point solver alone on caller-supplied minimal sets (normalised coordinates).
__global__ __launch_bounds__(ST) void five_point_debug_front_kernel(const double* __restrict__ x1,
                                                                    const double* __restrict__ x2, int count,
                                                                    double* __restrict__ recs,
                                                                    const int32_t* __restrict__ retry) {
    __shared__ double sA[200 * ST];
    const int h = blockIdx.x * ST + threadIdx.x;
    if (h >= count) return;
    double a[10], b[10];
    for (int i = 0; i < 10; ++i) {
        a[i] = x1[h * 10 + i];
        b[i] = x2[h * 10 + i];
    }
    const bool active = !retry || (retry[h] & 1);
    solve_front(active, a, b, sA, recs + (size_t)h * HYP_REC, retry ? 1 : 0);
}

__global__ __launch_bounds__(BACK_THREADS) void five_point_debug_back_kernel(int count, const double* __restrict__ recs,
                                                                             double* __restrict__ models,
                                                                             int32_t* __restrict__ nmodels, int literal) {
    __shared__ BackShared sh;
    const int h0 = blockIdx.x * HB;
    solve_back_block(sh, min(HB, count - h0), recs + (size_t)h0 * HYP_REC, models + (size_t)h0 * kMaxModels * 9,
                     nmodels + h0, literal);
}

// Hypotheses per round.  A round solves all of its hypotheses for every pair that is still
// running, so what a pair solves beyond its final iteration count is waste; but a round is also
// a chain of some ten launches, and small launches use the GPU badly.  Measured on batches of
// 512 pairs: eight rounds of 128 lose 10 % on a scene whose counts spread over 100..1000,
// {32, 96, 128, 256, 488} was the setting for a long time, merging its third and fourth round
// gains 4 % there and 3 % on a 100-pair sequence (fewer launches), finer rounds at the top
// ({32, 96, 256, 308, 308}) lose 1.5 %.  A call on a few pairs is nothing but its launch chain -
// some hundred five-point solves more or less do not show - so there the second and third round
// are one ({512, 488} measured slower on a pair that ends within 32 hypotheses: the kernels of a
// 512-hypothesis round take 0.3 ms however few pairs they serve).
static int round_size(int round, int batch) {
    static const int large[] = {32, 96, 384, 488}, small[] = {32, 480, 488, 0};
    return round < 4 ? (batch <= 32 ? small : large)[round] : 0;
}

int ransac_dev(vo_ctx* c, const float* d_p1, const float* d_p2, const int32_t* d_n, int n_max, int batch,
               const double K[4], double threshold, double confidence, int fixed_iters, vo_pair_result* d_res,
               double* d_E, uint8_t* d_mask_e, uint8_t* d_mask_pose, int32_t* d_good, int32_t* d_samples,
               int samples_cap, bool on_side, bool ask_after_first_round) {
    if (batch <= 0) return VO_OK;
    // on_side: the caller (the batched pipeline) has already ordered the side stream behind the
    // matches; everything below is enqueued there, one batch after the other, so the workspace
    // needs no further ordering.  Otherwise the work goes to the main stream, which first waits
    // for rounds of an earlier pipeline call that may still use the workspace.
    if (!on_side)
        if (int jrc = join_side(c)) return jrc;
    hipStream_t s = on_side ? c->side : c->stream;
    RParams P;
    P.ifx = 1.0 / K[0];
    P.ify = 1.0 / K[1];
    P.bx = -K[2] * P.ifx;
    P.by = -K[3] * P.ify;
    P.fx_mean = (K[0] + K[1]) / 2;
    P.threshold = threshold / P.fx_mean;
    P.t2 = (float)(P.threshold * P.threshold);
    P.t2_mid = 0.5 * ((double)P.t2 + (double)std::nextafter(P.t2, std::numeric_limits<float>::infinity()));
    P.confidence = confidence;
    P.fixed_iters = fixed_iters;
    // Largest round.  The adaptive loop works in rounds of at most 512 hypotheses (a pair stops at
    // the end of the round in which its bound is reached, so larger rounds solve more in vain).  A
    // fixed count has no early exit and its sample stream does not depend on any outcome: there
    // all hypotheses are drawn first and solved and scored in one round (in several if a batch
    // times the count exceeds 2^18 hypotheses, 600 MB of workspace), and only the bookkeeping walks
    // it in pieces of 512 - eight launches per stage of 4096 hypotheses become one.
    const int kChunk = UPD_MAX_HR;
    int HR = kChunk;
    if (fixed_iters > kChunk) {
        const long long cap = std::max<long long>(kChunk, (262144ll / batch) / kChunk * kChunk);
        HR = (int)std::min<long long>(((long long)fixed_iters + kChunk - 1) / kChunk * kChunk, cap);
    }
    // the minimal solver without refinement, second chart or duplicate drop, as OpenCV runs it
    const int literal = c->opt[VO_OPT_FIVE_POINT_LITERAL];
    int rc;
    int nm = std::max(n_max, 1);
    if ((rc = ensure(c, c->r_state, (size_t)batch * sizeof(RState)))) return rc;
    if ((rc = ensure(c, c->r_pts, (size_t)batch * nm * 4 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->r_ptsf, (size_t)batch * nm * 4 * sizeof(float)))) return rc;
    if ((rc = ensure(c, c->r_samples, (size_t)batch * HR * 5 * 4))) return rc;
    if ((rc = ensure(c, c->r_models, (size_t)batch * HR * kMaxModels * 9 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->r_nmodels, (size_t)batch * HR * 4))) return rc;
    if ((rc = ensure(c, c->r_models2, (size_t)batch * HR * kMaxModels * 9 * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->r_nmodels2, (size_t)batch * HR * 4))) return rc;
    if ((rc = ensure(c, c->r_retry, ((size_t)batch * HR + 1) * 4))) return rc;
    if ((rc = ensure(c, c->r_roots, (size_t)batch * HR * kMaxModels * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->r_nroots, (size_t)batch * HR * 4))) return rc;
    if ((rc = ensure(c, c->r_best, (size_t)batch * HR * HYP_REC * sizeof(double)))) return rc;
    if ((rc = ensure(c, c->r_counts, (size_t)batch * HR * kMaxModels * 4))) return rc;
    if ((rc = ensure(c, c->r_mask, (size_t)batch * (HR / kChunk + 1) * sizeof(FixedPart)))) return rc;   // per piece of a fixed-count round
    {
        static std::mutex jump_mutex;
        static bool jump_loaded[64] = {};
        std::lock_guard<std::mutex> lk(jump_mutex);
        if (!jump_loaded[c->device]) {
            unsigned long long t[64], g = 1;
            for (int j = 0; j < 64; ++j) {
                t[j] = g;
                for (int k = 0; k < UPD_CHUNK; ++k) g = (unsigned long long)(uint32_t)g * kMwcA + (uint32_t)(g >> 32);
            }
            VO_HIP(c, hipMemcpyToSymbol(HIP_SYMBOL(d_mwc_jump), t, sizeof(t)));
            jump_loaded[c->device] = true;
        }
    }
    RState* st = (RState*)c->r_state.p;
    double* m = (double*)c->r_pts.p;
    float* mf = (float*)c->r_ptsf.p;
    int32_t* samples = (int32_t*)c->r_samples.p;
    double* models = (double*)c->r_models.p;
    int32_t* nmodels = (int32_t*)c->r_nmodels.p;
    double* models2 = (double*)c->r_models2.p;
    int32_t* nmodels2 = (int32_t*)c->r_nmodels2.p;
    int32_t* retry_list = (int32_t*)c->r_retry.p;
    double* roots = (double*)c->r_roots.p;
    int32_t* nroots = (int32_t*)c->r_nroots.p;
    double* recs = (double*)c->r_best.p;
    int32_t* counts = (int32_t*)c->r_counts.p;
    FixedPart* parts = (FixedPart*)c->r_mask.p;

    t_begin_on(c, VO_T_RANSAC, s);
    if (n_max > 0)
        hipLaunchKernelGGL(normalize_kernel, dim3((n_max + 255) / 256, batch), dim3(256), 0, s, d_p1, d_p2, d_n, n_max, P,
                           m, mf);
    const int total = fixed_iters > 0 ? fixed_iters : kMaxIters;
    auto round_of = [&](int round, int h0) {   // hypotheses of a round that starts at h0
        if (h0 >= total) return 0;
        return std::min(fixed_iters > 0 ? HR : round_size(round, batch), total - h0);
    };
    auto update = [&](int h0, int hr, int arr_stride, int arr_off, int next, int out_off, bool last) {
        hipLaunchKernelGGL(update_kernel, dim3(batch), dim3(UPD_T), 0, s, st, batch, P, h0, hr, arr_stride, arr_off,
                           (const double*)models, (const int32_t*)nmodels, (const int32_t*)counts, samples, HR * 5, next,
                           out_off, last ? 1 : 0, d_samples, samples_cap, retry_list);
    };
    // sets [from, upto) of the round that starts at h0, beyond the first piece its predecessor drew
    auto draw_rest = [&](int h0, int hr, int from) {
        for (int d = from; d < hr; d += kChunk) update(h0 + d, 0, hr, 0, std::min(kChunk, hr - d), d, false);
    };
    // (init_kernel draws with one lane per pair: fine for the 32 sets of the adaptive loop's first
    // round, 170 us for 512 - a fixed-count round has all its pieces drawn by the workgroup routine)
    const bool init_draws = fixed_iters <= 0;
    hipLaunchKernelGGL(init_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, st, d_n, n_max, batch, P, samples, HR * 5,
                       init_draws ? std::min(round_of(0, 0), kChunk) : 0, retry_list);
    int h0 = 0;
    for (int round = 0; h0 < total; ++round) {
        const int hr = round_of(round, h0);
        if (hr <= 0) break;
        const int next = round_of(round + 1, h0 + hr);
        draw_rest(h0, hr, round == 0 && !init_draws ? 0 : kChunk);
        t_begin_on(c, VO_T_RANSAC_SOLVE, s);
        hipLaunchKernelGGL(solve_front_kernel, dim3((hr + ST - 1) / ST, batch), dim3(ST), 0, s, st, m, nm, samples, HR * 5,
                           h0, hr, recs);
// the root stage one wavefront per workgroup; VO_OPT_SOLVE_VARIANT = 1: one workgroup of 512
        // lanes per 64 hypotheses (the form the second-chart pass uses) - both give the same roots
        if (c->opt[VO_OPT_SOLVE_VARIANT] & 1)
            hipLaunchKernelGGL((solve_roots_kernel<BACK_THREADS, HB>), dim3((hr + HB - 1) / HB, batch), dim3(BACK_THREADS), 0,
                               s, st, hr, recs, roots, nroots);
        else
            hipLaunchKernelGGL(solve_roots_wave_kernel, dim3((hr + RW_H - 1) / RW_H, batch), dim3(RW_H * RW_L), 0, s, st, hr,
                               recs, roots, nroots);
        hipLaunchKernelGGL((solve_models_kernel<BACK_THREADS, HBM>), dim3((hr + HBM - 1) / HBM, batch), dim3(BACK_THREADS), 0,
                           s, st, hr, recs, (const double*)roots, (const int32_t*)nroots, models, nmodels,
                           literal ? nullptr : retry_list, literal);
        // Second chart for the few hypotheses with an odd number of models: grids that stride over
        // the list (its count is zeroed again by update_kernel).  (Running this pass on
        // a stream of its own beside the scoring of the first pass's models, and scoring what it
        // adds afterwards, measured no faster: the chain is short against the scoring kernel and
        // the two event hand-offs per round cost what the overlap gains.)
        if (c->opt[VO_OPT_SECOND_CHART] && !literal) {
            // (a workgroup per 64 list entries if every tenth hypothesis is on the list - more than
            // measured on any scene -, at least 64; workgroups beyond the list return at once)
            const int retry_front_wgs = std::min(8192, std::max(256, (int)(((long long)batch * hr / 10 + ST - 1) / ST)));
            const int retry_wgs = std::min(2048, std::max(64, (int)(((long long)batch * hr / 10 + HB - 1) / HB)));
            hipLaunchKernelGGL(retry_front_kernel, dim3(retry_front_wgs), dim3(ST), 0, s, m, nm, samples, HR * 5, hr,
                               (const int32_t*)retry_list, recs);
            hipLaunchKernelGGL(retry_back_kernel, dim3(retry_wgs), dim3(BACK_THREADS), 0, s, (const int32_t*)retry_list,
                               (const double*)recs, models2, nmodels2);
            hipLaunchKernelGGL(merge_models_kernel, dim3(16), dim3(256), 0, s, (const int32_t*)retry_list, 0, models, nmodels,
                               (const double*)models2, (const int32_t*)nmodels2);
        }
        t_end_on(c, VO_T_RANSAC_SOLVE, s);
        auto score = [&](int lo, int hi) {
            t_begin_on(c, VO_T_RANSAC_SCORE, s);
            hipLaunchKernelGGL(score_kernel, dim3((hi - lo + 3) / 4, batch), dim3(256), 0, s, st, m, (const float*)mf, nm, hr,
                               models, nmodels, P.t2, P.t2_mid, counts, lo, hi);
            t_end_on(c, VO_T_RANSAC_SCORE, s);
        };
        if (fixed_iters > 0 && hr > kChunk) {
            // A large fixed-count round: every piece of 512 finds its first maximum, one lane per
            // pair picks among them.  Scoring and picking go in stages - the first pieces, then
            // the rest - because the scoring kernel drops a hypothesis once it cannot beat what
            // the pair already has, and within one launch it has nothing to compare with.
            const int n_pieces = (hr + kChunk - 1) / kChunk;
            const int stage_end[2] = {1, n_pieces};
            for (int p_lo = 0, k = 0; p_lo < n_pieces; p_lo = stage_end[k++]) {
                const int p_hi = stage_end[k];
                score(p_lo * kChunk, std::min(p_hi * kChunk, hr));
                hipLaunchKernelGGL(fixed_scan_kernel, dim3(p_hi - p_lo, batch), dim3(FX_T), 0, s, (const RState*)st, h0, hr,
                                   (const int32_t*)nmodels, (const int32_t*)counts, parts, (const int32_t*)samples, HR * 5,
                                   d_samples, samples_cap, p_lo, n_pieces);
                hipLaunchKernelGGL(fixed_pick_kernel, dim3((batch + 63) / 64), dim3(64), 0, s, st, batch, h0, hr, n_pieces,
                                   p_lo, p_hi, (const FixedPart*)parts, (const double*)models, (const int32_t*)nmodels,
                                   (const int32_t*)counts, next <= 0 ? 1 : 0, retry_list);
            }
            // (a call that only draws fetches the next round's first piece)
            if (next > 0) update(h0 + hr, 0, hr, 0, std::min(next, kChunk), 0, false);
        } else {
            // the adaptive loop's replay (a round is at most 512 hypotheses), which also draws the
            // next round's sets
            score(0, hr);
            update(h0, hr, hr, 0, std::min(next, kChunk), 0, next <= 0);
        }
        h0 += hr;
        // A caller that waits for the result anyway (one pair from host points) and whose last call
        // ended within the first round asks here whether this one has, too: a 4-byte read-back
        // against the sixteen launches of two rounds that would find the pair finished.
        if (round == 0 && ask_after_first_round && batch == 1 && fixed_iters <= 0) {
            int done = 0;
            VO_HIP(c, hipMemcpyAsync(&done, (const char*)st + offsetof(RState, done), sizeof(int), hipMemcpyDeviceToHost, s));
            VO_HIP(c, hipStreamSynchronize(s));
            if (done) break;
        }
    }
    hipLaunchKernelGGL(finish_kernel, dim3(batch), dim3(4 * FIN_T), 0, s, st, m, nm, P, d_res, d_E, d_mask_e, d_mask_pose,
                       d_good);
    t_end_on(c, VO_T_RANSAC, s);
    if (on_side) {
        VO_HIP(c, hipEventRecord(c->ev_rdone, c->side));
        c->side_busy = true;
    }
    VO_HIP(c, hipGetLastError());
    return VO_OK;
}

}  // namespace vo

using namespace vo;

static int run_host_ransac(vo_ctx* c, const float* p1, const float* p2, int n, const double K[4], double threshold,
                           double confidence, int fixed_iters, vo_pair_result* res, double* E, uint8_t* mask_e,
                           uint8_t* mask_pose, int32_t* good2, int32_t* samples, int samples_cap) {
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    int nm = std::max(n, 1);
    if ((rc = ensure(c, c->m_p1, (size_t)nm * 8))) return rc;
    if ((rc = ensure(c, c->m_p2, (size_t)nm * 8))) return rc;
    size_t log_bytes = samples ? (size_t)samples_cap * 5 * 4 : 0;
    size_t res_bytes = sizeof(vo_pair_result) + 9 * 8 + 2 * (size_t)nm + 64 + log_bytes + 64;
    if ((rc = ensure(c, c->r_res, res_bytes))) return rc;
    uint8_t* base = (uint8_t*)c->r_res.p;
    vo_pair_result* d_res = (vo_pair_result*)base;
    double* d_E = (double*)(base + sizeof(vo_pair_result));
    int32_t* d_good = (int32_t*)(base + sizeof(vo_pair_result) + 72);
    int32_t* d_samples = samples ? (int32_t*)(base + sizeof(vo_pair_result) + 72 + 64) : nullptr;
    uint8_t* d_me = base + sizeof(vo_pair_result) + 72 + 64 + log_bytes;
    uint8_t* d_mp = d_me + nm;
    // the staging buffers and the workspace may still be in use by a pipeline call's tail on the
    // side stream
    if ((rc = join_side(c))) return rc;
    if (n > 0) {
        VO_HIP(c, hipMemcpyAsync(c->m_p1.p, p1, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
        VO_HIP(c, hipMemcpyAsync(c->m_p2.p, p2, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
    }
    if (d_samples) VO_HIP(c, hipMemsetAsync(d_samples, 0xFF, log_bytes, c->stream));
    if ((rc = ransac_dev(c, (const float*)c->m_p1.p, (const float*)c->m_p2.p, nullptr, n, 1, K, threshold, confidence,
                         fixed_iters, d_res, d_E, d_me, d_mp, d_good, d_samples, samples_cap, false, c->ransac_short_hint)))
        return rc;
    VO_HIP(c, hipMemcpyAsync(res, d_res, sizeof(vo_pair_result), hipMemcpyDeviceToHost, c->stream));
    if (E) VO_HIP(c, hipMemcpyAsync(E, d_E, 72, hipMemcpyDeviceToHost, c->stream));
    if (good2) VO_HIP(c, hipMemcpyAsync(good2, d_good, 8, hipMemcpyDeviceToHost, c->stream));
    if (mask_e && n > 0) VO_HIP(c, hipMemcpyAsync(mask_e, d_me, n, hipMemcpyDeviceToHost, c->stream));
    if (mask_pose && n > 0) VO_HIP(c, hipMemcpyAsync(mask_pose, d_mp, n, hipMemcpyDeviceToHost, c->stream));
    if (samples) VO_HIP(c, hipMemcpyAsync(samples, d_samples, log_bytes, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    // (what the next call from host points goes by: whether this one ended within the first round)
    c->ransac_short_hint = fixed_iters <= 0 && res->ransac_iters > 0 && res->ransac_iters <= round_size(0, 1);
    return VO_OK;
}

extern "C" int vo_estimate_motion(vo_ctx* c, const float* p1, const float* p2, int n, const double K[4],
                                  double threshold, double confidence, int fixed_iters, double R[9], double t[3],
                                  int* inliers, int* valid, uint8_t* mask) {
    if (!c) return VO_ERR_INVALID;
    if (!inliers || !valid || !R || !t || !K || n < 0 || (n > 0 && (!p1 || !p2)))
        return fail(c, VO_ERR_INVALID, "vo_estimate_motion: bad argument");
    *inliers = 0;
    *valid = 0;
    if (n < 5) return VO_OK;   // fewer than min_essential_points -> invalid, not an error
    vo_pair_result res;
    int rc = run_host_ransac(c, p1, p2, n, K, threshold, confidence, fixed_iters, &res, nullptr, nullptr, mask, nullptr,
                             nullptr, 0);
    if (rc) return rc;
    *valid = res.valid;
    *inliers = res.inliers;
    if (res.valid) {
        for (int k = 0; k < 9; ++k) R[k] = res.R[k];
        for (int k = 0; k < 3; ++k) t[k] = res.t[k];
    }
    return VO_OK;
}

extern "C" int vo_find_essential(vo_ctx* c, const float* p1, const float* p2, int n, const double K[4],
                                 double threshold, double confidence, int fixed_iters, double E[9], uint8_t* mask,
                                 int* good, int* iters_run, int32_t* samples) {
    if (!c) return VO_ERR_INVALID;
    if (!good || !E || !K || n < 0 || (n > 0 && (!p1 || !p2)))
        return fail(c, VO_ERR_INVALID, "vo_find_essential: bad argument");
    *good = 0;
    if (iters_run) *iters_run = 0;
    if (n < 5) return VO_OK;
    vo_pair_result res;
    int32_t g2[2] = {0, 0};
    int cap = std::max(1000, fixed_iters);
    int rc = run_host_ransac(c, p1, p2, n, K, threshold, confidence, fixed_iters, &res, E, mask, nullptr, g2, samples,
                             cap);
    if (rc) return rc;
    *good = g2[0];
    if (iters_run) *iters_run = g2[1];
    return VO_OK;
}

extern "C" int vo_debug_five_point(vo_ctx* c, const double* x1, const double* x2, int count, double* Es,
                                   int32_t* n_models) {
    if (!c) return VO_ERR_INVALID;
    if (count <= 0 || !x1 || !x2 || !Es || !n_models) return fail(c, VO_ERR_INVALID, "vo_debug_five_point: bad argument");
    VO_HIP(c, hipSetDevice(c->device));
    int rc;
    if ((rc = join_side(c))) return rc;
    size_t in_b = (size_t)count * 10 * 8, out_b = (size_t)count * kMaxModels * 9 * 8;
    if ((rc = ensure(c, c->r_pts, 2 * in_b))) return rc;
    if ((rc = ensure(c, c->r_models, out_b))) return rc;
    if ((rc = ensure(c, c->r_nmodels, (size_t)count * 4))) return rc;
    double* d_in = (double*)c->r_pts.p;
    VO_HIP(c, hipMemcpyAsync(d_in, x1, in_b, hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemcpyAsync(d_in + (size_t)count * 10, x2, in_b, hipMemcpyHostToDevice, c->stream));
    VO_HIP(c, hipMemsetAsync(c->r_models.p, 0, out_b, c->stream));
    if ((rc = ensure(c, c->r_best, (size_t)count * HYP_REC * 8))) return rc;
    if ((rc = ensure(c, c->r_models2, out_b))) return rc;
    if ((rc = ensure(c, c->r_nmodels2, (size_t)count * 4))) return rc;
    hipLaunchKernelGGL(five_point_debug_front_kernel, dim3((count + ST - 1) / ST), dim3(ST), 0, c->stream, d_in,
                       d_in + (size_t)count * 10, count, (double*)c->r_best.p, (const int32_t*)nullptr);
    const int literal = c->opt[VO_OPT_FIVE_POINT_LITERAL];
    hipLaunchKernelGGL(five_point_debug_back_kernel, dim3((count + HB - 1) / HB), dim3(BACK_THREADS), 0, c->stream, count,
                       (const double*)c->r_best.p, (double*)c->r_models.p, (int32_t*)c->r_nmodels.p, literal);
    // the second-chart pass for odd model counts, as in the RANSAC rounds
    if (!literal && c->opt[VO_OPT_SECOND_CHART]) {
        hipLaunchKernelGGL(five_point_debug_front_kernel, dim3((count + ST - 1) / ST), dim3(ST), 0, c->stream, d_in,
                           d_in + (size_t)count * 10, count, (double*)c->r_best.p, (const int32_t*)c->r_nmodels.p);
        hipLaunchKernelGGL(five_point_debug_back_kernel, dim3((count + HB - 1) / HB), dim3(BACK_THREADS), 0, c->stream,
                           count, (const double*)c->r_best.p, (double*)c->r_models2.p, (int32_t*)c->r_nmodels2.p, 0);
        hipLaunchKernelGGL(merge_models_kernel, dim3((count + 255) / 256), dim3(256), 0, c->stream,
                           (const int32_t*)nullptr, count, (double*)c->r_models.p, (int32_t*)c->r_nmodels.p,
                           (const double*)c->r_models2.p, (const int32_t*)c->r_nmodels2.p);
    }
    VO_HIP(c, hipGetLastError());
    VO_HIP(c, hipMemcpyAsync(Es, c->r_models.p, out_b, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipMemcpyAsync(n_models, c->r_nmodels.p, (size_t)count * 4, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipStreamSynchronize(c->stream));
    return VO_OK;
}
