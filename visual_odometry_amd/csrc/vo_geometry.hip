// Essential-matrix RANSAC, pose recovery and DLT triangulation for gfx950.
//
// Replaces, inside optRelativePose / triangPoints of the reference
// (include/visual_odometry/camera_utility.hpp:633-685, 288-351):
//   findEssentialMat(p1,p2,K,RANSAC,0.99,2.0,mask) :642, extract_Inlier :655, recoverPose :667,
//   triangulatePoints :320, my_convertFromHom :326, reproject_error :330-331, filter_convertWP :348.
//
//   k_geo_prep        : pixel -> normalised fp64 coordinates, per-pair state reset
//   k_ransac_tuples   : the registrator's 5-tuples from the cv::RNG stream (raw 32-bit outputs are
//                       count-independent and precomputed on the host; modulo, duplicate rejection
//                       and stream alignment happen here, speculatively 64 tuples at a time)
//   k_ransac_hyp      : ONE WAVEFRONT PER HYPOTHESIS: null space of the 5x9 system, 10x20
//                       Gauss-Jordan, degree-10 resultant, Durand-Kerner roots (one lane per root),
//                       back-substitution (one lane per real root), then Sampson scoring of every
//                       candidate E against all points (lanes over points, ballot counts)
//   k_ransac_scan     : in-order scan reproducing the sequential "goodCount > max(best,4)" /
//                       RANSACUpdateNumIters loop; hypotheses past the live bound exit early
//   k_ransac_finish   : inlier mask of the winner, ordered inlier compaction, gates
//   k_pose_decompose / k_pose_cheirality / k_pose_select : recoverPose
//   k_triang / k_triang_filter : DLT per point, reprojection filter, frame change, 3-sigma filter
// All arithmetic is fp64 (float only where OpenCV stores float), compiled with -ffp-contract=off.
#include <cfloat>
#include <cmath>

#include "vo_common.hpp"

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
// lane k <- lane (k + N) mod 16 of its aligned 16-lane row: one DPP move per 32-bit half (no LDS crossbar round trip)
template <int N>
__device__ __forceinline__ double row_ror(double v) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x120 + N, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x120 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
// Sum over an aligned 16-lane group, the same in every lane.  Butterfly k ^ 8, k ^ 4, k ^ 2, k ^ 1 written as row rotations:
// after the step with distance d the values have period d, so lane (k + d/2) mod 16 holds what lane k ^ (d/2) holds - the
// operands of every add are those of the xor butterfly (bit-identical, additions commute).
__device__ __forceinline__ double sum16(double v) {
    v += row_ror<8>(v);
    v += row_ror<4>(v);
    v += row_ror<2>(v);
    v += row_ror<1>(v);
    return v;
}
// value of lane `src` (compile-time constant after unrolling) in every lane: two v_readlane, result in scalar registers
__device__ __forceinline__ double lane_bcast(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One-sided Jacobi SVD of the rows of At (OpenCV JacobiSVDImpl_<double>), N <= 4,
// fully unrolled so that everything stays in registers.  On return rows of At are normalised left
// vectors, W the singular values (descending) and rows of V the right vectors.
template <int N>
__device__ void jacobi_rows(double (&At)[N][N], double (&W)[N], double (&V)[N][N]) {
    const double eps = DBL_EPSILON * 10, minval = DBL_MIN;
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i][k] * At[i][k];
        W[i] = sd;
#pragma unroll
        for (int k = 0; k < N; k++) V[i][k] = (i == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; iter++) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < N - 1; i++)
#pragma unroll
            for (int j = i + 1; j < N; j++) {
                double a = W[i], p = 0, b = W[j];
#pragma unroll
                for (int k = 0; k < N; k++) p += At[i][k] * At[j][k];
                if (fabs(p) <= eps * sqrt(a * b)) continue;
                p *= 2;
                const double beta = a - b, gamma = hypot(p, beta);
                double c, s;
                if (beta < 0) {
                    const double delta = (gamma - beta) * 0.5;
                    s = sqrt(delta / gamma);
                    c = p / (gamma * s * 2);
                } else {
                    c = sqrt((gamma + beta) / (gamma * 2));
                    s = p / (gamma * c * 2);
                }
                a = b = 0;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * At[i][k] + s * At[j][k];
                    const double t1 = -s * At[i][k] + c * At[j][k];
                    At[i][k] = t0;
                    At[j][k] = t1;
                    a += t0 * t0;
                    b += t1 * t1;
                }
                W[i] = a;
                W[j] = b;
                changed = true;
#pragma unroll
                for (int k = 0; k < N; k++) {
                    const double t0 = c * V[i][k] + s * V[j][k];
                    const double t1 = -s * V[i][k] + c * V[j][k];
                    V[i][k] = t0;
                    V[j][k] = t1;
                }
            }
        if (!changed) break;
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < N; k++) sd += At[i][k] * At[i][k];
        W[i] = sqrt(sd);
    }
    // selection sort, descending (same swaps as the reference loop)
#pragma unroll
    for (int i = 0; i < N - 1; i++) {
        int j = i;
#pragma unroll
        for (int k = i + 1; k < N; k++) {
            double wj = W[i];
#pragma unroll
            for (int q = i + 1; q < N; q++)
                if (q == j) wj = W[q];
            if (wj < W[k]) j = k;
        }
#pragma unroll
        for (int k = i + 1; k < N; k++)
            if (k == j) {
                double tw = W[i];
                W[i] = W[k];
                W[k] = tw;
#pragma unroll
                for (int q = 0; q < N; q++) {
                    double t = At[i][q];
                    At[i][q] = At[k][q];
                    At[k][q] = t;
                    t = V[i][q];
                    V[i][q] = V[k][q];
                    V[k][q] = t;
                }
            }
    }
#pragma unroll
    for (int i = 0; i < N; i++) {
        const double s = W[i] > minval ? 1 / W[i] : 0.;
#pragma unroll
        for (int k = 0; k < N; k++) At[i][k] *= s;
    }
}

// cv::triangulatePoints for one point (A.11): X = last row of Vt of the 4x4 system
__device__ void dlt_point(const double* __restrict__ P0, const double* __restrict__ P1, double x0, double y0, double x1,
                          double y1, double X[4]) {
    double At[4][4], W[4], V[4][4];  // At rows = columns of A
#pragma unroll
    for (int k = 0; k < 4; k++) {
        At[k][0] = x0 * P0[8 + k] - P0[k];
        At[k][1] = y0 * P0[8 + k] - P0[4 + k];
        At[k][2] = x1 * P1[8 + k] - P1[k];
        At[k][3] = y1 * P1[8 + k] - P1[4 + k];
    }
    jacobi_rows<4>(At, W, V);
#pragma unroll
    for (int k = 0; k < 4; k++) X[k] = V[3][k];
}

// EMEstimatorCallback::computeError for one point (Sampson distance, stored as float)
__device__ __forceinline__ float sampson_err(const double E[9], double ax, double ay, double bx, double by) {
    const double Ex0 = E[0] * ax + E[1] * ay + E[2] * 1.;
    const double Ex1 = E[3] * ax + E[4] * ay + E[5] * 1.;
    const double Ex2 = E[6] * ax + E[7] * ay + E[8] * 1.;
    const double Et0 = E[0] * bx + E[3] * by + E[6] * 1.;
    const double Et1 = E[1] * bx + E[4] * by + E[7] * 1.;
    const double x2tEx1 = bx * Ex0 + by * Ex1 + 1. * Ex2;
    const double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    return (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
}

// findInliers' test (float)(n^2 / d) <= thr for one point WITHOUT the division in all but the rare borderline case: with
// T = (double)thr, n^2 <= T (1 - 1e-6) d puts the quotient a millionth below T (the float rounding moves it by 6e-8 at most, the
// double division by 1e-16), n^2 >= T (1 + 1e-6) d a millionth above it; only a point inside that band - or a degenerate
// d = 0, where the reference compares a NaN or an infinity - takes the division.  Same verdict as sampson_err(...) <= thr.
__device__ __forceinline__ bool sampson_inlier(const double E[9], double ax, double ay, double bx, double by, float thr,
                                               double thr_lo, double thr_hi) {
    const double Ex0 = E[0] * ax + E[1] * ay + E[2] * 1.;
    const double Ex1 = E[3] * ax + E[4] * ay + E[5] * 1.;
    const double Ex2 = E[6] * ax + E[7] * ay + E[8] * 1.;
    const double Et0 = E[0] * bx + E[3] * by + E[6] * 1.;
    const double Et1 = E[1] * bx + E[4] * by + E[7] * 1.;
    const double x2tEx1 = bx * Ex0 + by * Ex1 + 1. * Ex2;
    const double a = Ex0 * Ex0, b = Ex1 * Ex1, c = Et0 * Et0, d = Et1 * Et1;
    const double num = x2tEx1 * x2tEx1, den = a + b + c + d;
    const bool sure_in = num <= thr_lo * den && den > 0, sure_out = num >= thr_hi * den && den > 0;
    bool in = sure_in;
    if (!(sure_in || sure_out)) in = (float)(num / den) <= thr;
    return in;
}

__device__ int d_ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters) {
    p = fmax(p, 0.);
    p = fmin(p, 1.);
    ep = fmax(ep, 0.);
    ep = fmin(ep, 1.);
    double num = fmax(1. - p, DBL_MIN);
    double denom = 1. - pow(1. - ep, (double)modelPoints);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= maxIters * (-denom) ? maxIters : __double2int_rn(num / denom);
}

// ------------------------------------------------------------------------------------------------
// preparation
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_geo_prep(const float* __restrict__ p1, const float* __restrict__ p2, int match_cap,
                                                  double* __restrict__ m1, double* __restrict__ m2,
                                                  PairState* __restrict__ st, int max_iters, double thr_px, int reset_ransac) {
    const int pair = blockIdx.y;
    PairState& S = st[pair];
    const int n = S.n_used;
    const double fx = S.K[0], fy = S.K[4], cx = S.K[2], cy = S.K[5];  // skew ignored (A.8)
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const size_t o = ((size_t)pair * match_cap + i) * 2;
        m1[o] = ((double)p1[o] - cx) / fx;
        m1[o + 1] = ((double)p1[o + 1] - cy) / fy;
        m2[o] = ((double)p2[o] - cx) / fx;
        m2[o + 1] = ((double)p2[o + 1] - cy) / fy;
    }
    if (reset_ransac && blockIdx.x == 0 && threadIdx.x == 0) {
        const double threshold = thr_px / ((fx + fy) / 2);  // A.8
        S.thr_sq = (float)(threshold * threshold);
        S.niters = n < 5 ? 0 : (n == 5 ? 1 : max(max_iters, 1));
        S.best_count = 0;
        S.best_hyp = -1;
        S.best_model = 0;
        S.scanned = 0;
        S.n_inl = 0;
        S.good[0] = S.good[1] = S.good[2] = S.good[3] = 0;
        S.n_good = 0;
        S.success = 0;
        S.status = n < 5 ? VO_ERR_TOO_FEW_POINTS : VO_OK;
        S.n_pre = 0;
        S.n_world = 0;
        S.iters_run = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// cv::RNG sample stream -> 5-tuples.  raw[] = successive RNG::next() outputs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ransac_tuples(const uint32_t* __restrict__ raw, int raw_len,
                                                      PairState* __restrict__ st, int max_hyp, int* __restrict__ tuples) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    PairState& S = st[pair];
    const int n = S.n_used;
    int* out = tuples + (size_t)pair * max_hyp * 5;
    const int T = min(S.niters, max_hyp);
    if (n < 5 || T <= 0) return;
    if (n == 5) {  // count == modelPoints: the kernel runs once on the points as they are
        if (lane < 5) out[lane] = lane;
        return;
    }
    const unsigned cnt = (unsigned)n;
    int t = 0, pos = 0;
    while (t < T) {
        const int start = pos + 5 * lane;
        bool dup = false, oob = (start + 5 > raw_len);
        int d[5] = {0, 0, 0, 0, 0};
        if (!oob) {
#pragma unroll
            for (int q = 0; q < 5; q++) d[q] = (int)(raw[start + q] % cnt);
            dup = (d[0] == d[1]) || (d[0] == d[2]) || (d[0] == d[3]) || (d[0] == d[4]) || (d[1] == d[2]) ||
                  (d[1] == d[3]) || (d[1] == d[4]) || (d[2] == d[3]) || (d[2] == d[4]) || (d[3] == d[4]);
        }
        const unsigned long long m = __ballot(dup || oob);
        const int first = m ? (int)__ffsll((long long)m) - 1 : 64;
        const int adv = min(first, T - t);
        if (lane < adv) {
#pragma unroll
            for (int q = 0; q < 5; q++) out[(size_t)(t + lane) * 5 + q] = d[q];
        }
        t += adv;
        pos += 5 * adv;
        if (t < T && first < 64) {
            // the tuple at `pos` rejects at least one draw (or the stream ran out): walk it in order
            int idx[5] = {0, 0, 0, 0, 0};
            int i = 0;
            bool ran_out = false;
            while (i < 5) {
                if (pos >= raw_len) {
                    ran_out = true;
                    break;
                }
                const int v = (int)(raw[pos++] % cnt);
                bool seen = false;
#pragma unroll
                for (int q = 0; q < 5; q++) seen = seen || (q < i && idx[q] == v);
                if (!seen) {
#pragma unroll
                    for (int q = 0; q < 5; q++)
                        if (q == i) idx[q] = v;
                    i++;
                }
            }
            if (ran_out) {  // cannot happen with the stream length the host provides; stop cleanly
                if (lane == 0) S.niters = t;
                break;
            }
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 5; q++) out[(size_t)t * 5 + q] = idx[q];
            }
            t++;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// five-point solver + scoring: one wavefront per hypothesis
// ------------------------------------------------------------------------------------------------
// monomial order of the 10x20 system (A.9 step 3)
//   0:x^3 1:y^3 2:x^2y 3:xy^2 4:x^2z 5:x^2 6:y^2z 7:y^2 8:xyz 9:xy | 10:xz^2 11:xz 12:x 13:yz^2 14:yz 15:y
//   16:z^3 17:z^2 18:z 19:1
// linear polys: [x, y, z, 1]; quadratic polys: [x^2, y^2, z^2, xy, xz, yz, x, y, z, 1]
__device__ __forceinline__ void lin_mul_acc(const double a[4], const double b[4], double out[10], double sgn) {
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            // static table lookups fold at compile time after unrolling
            const int tbl[4][4] = {{0, 3, 4, 6}, {3, 1, 5, 7}, {4, 5, 2, 8}, {6, 7, 8, 9}};
            out[tbl[i][j]] += sgn * (a[i] * b[j]);
        }
}
__device__ __forceinline__ void quad_lin_mul_acc(const double a[10], const double b[4], double out[20], double sgn) {
#pragma unroll
    for (int i = 0; i < 10; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int tbl[10][4] = {{0, 2, 4, 5},   {3, 1, 6, 7},   {10, 13, 16, 17}, {2, 3, 8, 9},    {4, 8, 10, 11},
                                    {8, 6, 13, 14}, {5, 9, 11, 12}, {9, 7, 14, 15},   {11, 14, 17, 18}, {12, 15, 18, 19}};
            out[tbl[i][j]] += sgn * (a[i] * b[j]);
        }
}

struct HypShared {
    double At[9][9];     // rows 0..4: orthonormal basis of the row space of Q, rows 5..8: null space
    double Elin[9][4];   // E entry e as a linear polynomial: coefficients of x, y, z, 1
    double F[9][10];     // 2*E*E^T - tr*I entries (quadratic polynomials)
    double A[10][20];
    double fcol[10];
    double b[3][13];
    double term[3][11];
    double c[11];
    double E[kMaxModels][9];
    double zroot[10];
    int valid[10];
    int cnt[kMaxModels];
    int piv;
    int nm;
};

#ifdef VO_RANSAC_TIMING
__device__ unsigned long long g_rt[16];
#define RT_MARK(i) do { if (lane == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_rt[i], t_ - rt_prev); rt_prev = t_; } } while (0)
extern "C" int vo_debug_ransac_timing(unsigned long long* out, int reset) {
    if (out) hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rt), sizeof(g_rt));
    if (reset) { unsigned long long z[16] = {}; hipMemcpyToSymbol(HIP_SYMBOL(g_rt), z, sizeof(z)); }
    return 0;
}
#else
#define RT_MARK(i)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_ransac_hyp(const double* __restrict__ m1_base, const double* __restrict__ m2_base,
                                                   int match_cap, const int* __restrict__ tuples_base, int max_hyp,
                                                   int h0, int h1, const PairState* __restrict__ st, const int* __restrict__ signs,
                                                   double* __restrict__ models_base, int* __restrict__ nmodels_base,
                                                   int* __restrict__ counts_base) {
    __shared__ HypShared sh;
    const int pair = blockIdx.y;
    const int h = h0 + blockIdx.x;
    const int lane = threadIdx.x;
    if (h >= h1) return;
    int n = 0;
    float thr_sq = 0.f;
    if (st) {
        if (h >= st[pair].niters) return;  // past the live RANSAC bound: nothing to do
        n = st[pair].n_used;
        thr_sq = st[pair].thr_sq;
    }
    const double* __restrict__ m1 = m1_base + (size_t)pair * match_cap * 2;
    const double* __restrict__ m2 = m2_base + (size_t)pair * match_cap * 2;
    const int* tup = tuples_base + ((size_t)pair * max_hyp + h) * 5;
    double* models = models_base + ((size_t)pair * max_hyp + h) * kMaxModels * 9;
    int* counts = counts_base ? counts_base + ((size_t)pair * max_hyp + h) * kMaxModels : nullptr;

#ifdef VO_RANSAC_TIMING
    unsigned long long rt_prev = wall_clock64(), rt_iters = 0;
    if (lane == 0) atomicAdd(&g_rt[15], 1ull);
#endif
    // ---------------------------------------------------------------- (a) Q (5x9), lanes 0..8 hold column k
    const int k = lane & 15;  // column index within the 16-lane group; all four groups do the same work
    double a[9];              // a[i] = At[i][k]
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const int id = tup[i];
        const double x1 = m1[2 * id], y1 = m1[2 * id + 1], x2 = m2[2 * id], y2 = m2[2 * id + 1];
        // row of Q for vec(E) row-major and x2^T E x1 = 0
        const double q[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1.0};
        double v = 0;
#pragma unroll
        for (int c = 0; c < 9; c++)
            if (c == k) v = q[c];
        a[i] = (k < 9) ? v : 0.0;
    }
    // ---------------------------------------------------------------- (b) orthonormal row space (MGS, twice)
#pragma unroll
    for (int i = 0; i < 5; i++) {
#pragma unroll
        for (int rep = 0; rep < 2; rep++)
#pragma unroll
            for (int j = 0; j < i; j++) {
                const double d = sum16(a[i] * a[j]);
                a[i] -= d * a[j];
            }
        const double nn = sqrt(sum16(a[i] * a[i]));
        a[i] *= (nn > DBL_MIN ? 1 / nn : 0.);
    }
    // null space: OpenCV's JacobiSVD completes Vt by projecting fixed +-1/9 sign vectors (RNG
    // 0x12345678) out of the rows found so far; the projector does not depend on the basis chosen
    // for the row space, so the same four vectors come out here.
#pragma unroll
    for (int i = 5; i < 9; i++) {
        a[i] = (k < 9) ? (signs[(i - 5) * 9 + (k < 9 ? k : 0)] ? 1. / 9 : -1. / 9) : 0.0;
#pragma unroll
        for (int rep = 0; rep < 2; rep++)
#pragma unroll
            for (int j = 0; j < i; j++) {
                const double d = sum16(a[i] * a[j]);
                a[i] -= d * a[j];
                const double asum = sum16(fabs(a[i]));
                a[i] *= (asum > DBL_EPSILON * 1000 ? 1 / asum : 0.);
            }
        const double nn = sqrt(sum16(a[i] * a[i]));
        a[i] *= (nn > DBL_MIN ? 1 / nn : 0.);
    }
    if (lane < 9) {
#pragma unroll
        for (int i = 0; i < 9; i++) sh.At[i][lane] = a[i];
#pragma unroll
        for (int v = 0; v < 4; v++) sh.Elin[lane][v] = a[5 + v];  // E = x E1 + y E2 + z E3 + E4
    }
    __syncthreads();
    RT_MARK(0);

    // ---------------------------------------------------------------- (c) constraint polynomials
    // F = 2 E E^T - tr(E E^T) I  (symmetric, quadratic entries); lanes 0..5 each build one EEt entry
    if (lane < 6) {
        const int fi[6] = {0, 0, 0, 1, 1, 2}, fj[6] = {0, 1, 2, 1, 2, 2};
        int i = 0, j = 0;
#pragma unroll
        for (int q = 0; q < 6; q++)
            if (q == lane) {
                i = fi[q];
                j = fj[q];
            }
        double acc[10];
#pragma unroll
        for (int q = 0; q < 10; q++) acc[q] = 0;
#pragma unroll
        for (int kk = 0; kk < 3; kk++) {
            double ea[4], eb[4];
#pragma unroll
            for (int v = 0; v < 4; v++) {
                ea[v] = sh.Elin[3 * i + kk][v];
                eb[v] = sh.Elin[3 * j + kk][v];
            }
            lin_mul_acc(ea, eb, acc, 1.0);
        }
#pragma unroll
        for (int q = 0; q < 10; q++) {
            sh.F[3 * i + j][q] = acc[q];  // EEt for now
            sh.F[3 * j + i][q] = acc[q];
        }
    }
    __syncthreads();
    {
        // trace and F = 2 EEt - tr I: lanes 0..8 own one entry each
        double f[10];
        const int e = lane < 9 ? lane : 0;
#pragma unroll
        for (int q = 0; q < 10; q++) {
            const double tr = sh.F[0][q] + sh.F[4][q] + sh.F[8][q];
            f[q] = 2.0 * sh.F[e][q];
            if (e == 0 || e == 4 || e == 8) f[q] -= tr;
        }
        __syncthreads();  // everyone has read EEt before it is overwritten
        if (lane < 9) {
#pragma unroll
            for (int q = 0; q < 10; q++) sh.F[e][q] = f[q];
        }
    }
    __syncthreads();
    {
        double row[20];
#pragma unroll
        for (int q = 0; q < 20; q++) row[q] = 0;
        if (lane < 9) {
            // Lambda_ij = sum_k F_ik E_kj
            const int i = lane / 3, j = lane - 3 * (lane / 3);
#pragma unroll
            for (int kk = 0; kk < 3; kk++) {
                double f[10], e[4];
#pragma unroll
                for (int q = 0; q < 10; q++) f[q] = sh.F[3 * i + kk][q];
#pragma unroll
                for (int v = 0; v < 4; v++) e[v] = sh.Elin[3 * kk + j][v];
                quad_lin_mul_acc(f, e, row, 1.0);
            }
        } else if (lane == 9) {
            // det(E) = E00 (E11 E22 - E12 E21) - E01 (E10 E22 - E12 E20) + E02 (E10 E21 - E11 E20)
            const int ta[3] = {0, 1, 2};
            const int m0a[3] = {4, 3, 3}, m0b[3] = {8, 8, 7}, m1a[3] = {5, 5, 4}, m1b[3] = {7, 6, 6};
            const double sg[3] = {1.0, -1.0, 1.0};
#pragma unroll
            for (int t = 0; t < 3; t++) {
                double qd[10];
#pragma unroll
                for (int q = 0; q < 10; q++) qd[q] = 0;
                double u[4], v[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u[q] = sh.Elin[m0a[t]][q];
                    v[q] = sh.Elin[m0b[t]][q];
                }
                lin_mul_acc(u, v, qd, 1.0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    u[q] = sh.Elin[m1a[t]][q];
                    v[q] = sh.Elin[m1b[t]][q];
                }
                lin_mul_acc(u, v, qd, -1.0);
#pragma unroll
                for (int q = 0; q < 4; q++) u[q] = sh.Elin[ta[t]][q];
                quad_lin_mul_acc(qd, u, row, sg[t]);
            }
        }
        if (lane < 10) {
            const int r = lane == 9 ? 0 : lane + 1;  // row 0 = det, rows 1..9 = Lambda (oracle order)
#pragma unroll
            for (int q = 0; q < 20; q++) sh.A[r][q] = row[q];
        }
    }
    __syncthreads();
    RT_MARK(1);

    // ---------------------------------------------------------------- (d) Gauss-Jordan, partial pivoting
    bool singular = false;
    for (int col = 0; col < 10; col++) {
        if (lane == 0) {
            int piv = col;
            double best = fabs(sh.A[col][col]);
            for (int r = col + 1; r < 10; r++) {
                const double v = fabs(sh.A[r][col]);
                if (v > best) {
                    best = v;
                    piv = r;
                }
            }
            sh.piv = (best == 0) ? -1 : piv;
        }
        __syncthreads();
        const int piv = sh.piv;
        if (piv < 0) {
            singular = true;
            break;
        }
        if (lane < 20 && piv != col) {
            const double t = sh.A[piv][lane];
            sh.A[piv][lane] = sh.A[col][lane];
            sh.A[col][lane] = t;
        }
        __syncthreads();
        const double inv = 1.0 / sh.A[col][col];
        __syncthreads();
        if (lane < 20) sh.A[col][lane] *= inv;
        if (lane >= 32 && lane < 42) sh.fcol[lane - 32] = sh.A[lane - 32][col];
        __syncthreads();
        for (int e = lane; e < 200; e += 64) {
            const int r = e / 20, c = e - 20 * r;
            if (r != col) {
                const double f = sh.fcol[r];
                if (f != 0) sh.A[r][c] -= f * sh.A[col][c];
            }
        }
        __syncthreads();
    }
    if (singular) {
        if (lane == 0) nmodels_base[(size_t)pair * max_hyp + h] = 0;
        return;
    }

    RT_MARK(2);
    // ---------------------------------------------------------------- (e) B(z) and its determinant
    if (lane < 3) {
        const double* ar1 = &sh.A[4 + 2 * lane][10];
        const double* ar2 = &sh.A[5 + 2 * lane][10];
        double row1[13], row2[13];
#pragma unroll
        for (int q = 0; q < 13; q++) row1[q] = row2[q] = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            row1[1 + q] = ar1[q];
            row1[5 + q] = ar1[3 + q];
            row2[0 + q] = ar2[q];
            row2[4 + q] = ar2[3 + q];
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            row1[9 + q] = ar1[6 + q];
            row2[8 + q] = ar2[6 + q];
        }
#pragma unroll
        for (int q = 0; q < 13; q++) sh.b[lane][q] = row1[q] - row2[q];
    }
    __syncthreads();
    if (lane < 3) {
        // term(lane) = sign * f * (g1 h1 - g2 h2), polynomials in ascending powers of z
        // lane 0: p0 (q1 r2 - q2 r1);  lane 1: -q0 (p1 r2 - p2 r1);  lane 2: r0 (p1 q2 - p2 q1)
        double P[3][5], Q[3][5], R[3][5];
#pragma unroll
        for (int i = 0; i < 3; i++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                P[i][q] = sh.b[i][3 - q];
                Q[i][q] = sh.b[i][7 - q];
            }
            P[i][4] = 0;
            Q[i][4] = 0;
#pragma unroll
            for (int q = 0; q < 5; q++) R[i][q] = sh.b[i][12 - q];
        }
        double f[5], g1[5], h1[5], g2[5], h2[5];
        int df = 3, dg = 3, dh = 4;
        double sgn = 1.0;
#pragma unroll
        for (int q = 0; q < 5; q++) {
            if (lane == 0) {
                f[q] = P[0][q];
                g1[q] = Q[1][q];
                h1[q] = R[2][q];
                g2[q] = Q[2][q];
                h2[q] = R[1][q];
            } else if (lane == 1) {
                f[q] = Q[0][q];
                g1[q] = P[1][q];
                h1[q] = R[2][q];
                g2[q] = P[2][q];
                h2[q] = R[1][q];
            } else {
                f[q] = R[0][q];
                g1[q] = P[1][q];
                h1[q] = Q[2][q];
                g2[q] = P[2][q];
                h2[q] = Q[1][q];
            }
        }
        if (lane == 1) sgn = -1.0;
        if (lane == 2) {
            df = 4;
            dh = 3;
        }
        double t1[9], t2[9], tt[9], out[13];
#pragma unroll
        for (int q = 0; q < 9; q++) t1[q] = t2[q] = 0;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (i <= dg && j <= dh) {
                    t1[i + j] += g1[i] * h1[j];
                    t2[i + j] += g2[i] * h2[j];
                }
            }
#pragma unroll
        for (int q = 0; q < 9; q++) tt[q] = t1[q] - t2[q];
#pragma unroll
        for (int q = 0; q < 13; q++) out[q] = 0;
#pragma unroll
        for (int i = 0; i < 5; i++)
#pragma unroll
            for (int j = 0; j < 9; j++) {
                if (i <= df && j <= dg + dh) out[i + j] += f[i] * tt[j];
            }
#pragma unroll
        for (int q = 0; q < 11; q++) sh.term[lane][q] = sgn * out[q];
    }
    __syncthreads();
    if (lane < 11) {
        double c = 0;
        c += sh.term[0][lane];
        c += sh.term[1][lane];
        c += sh.term[2][lane];
        sh.c[lane] = c;
    }
    __syncthreads();

    RT_MARK(3);
    // ---------------------------------------------------------------- (f) roots: Durand-Kerner, lane per root
    // The roots live in registers (lane j holds root j) and reach the other lanes through v_readlane; the coefficients are
    // the same in every lane.  One iteration is a single pass with no LDS traffic and no barrier (it used to be two barriers
    // and 2 * deg LDS reads per lane: 1.4 us per iteration, 60 % of a hypothesis).  Every lane reads the OLD roots of all
    // lanes before any is updated, and the operations and their order are unchanged, so the roots are bit for bit the same.
    int deg = 10;
    for (; deg > 1; deg--)
        if (fabs(sh.c[deg]) > DBL_EPSILON) break;
    double cf[10];  // cf[j] = c[deg - j - 1]: the coefficient Horner's step j adds
#pragma unroll
    for (int j = 0; j < 10; j++) cf[j] = lane_bcast(sh.c[max(deg - j - 1, 0)], 0);  // uniform: scalar registers
    const double cdeg = lane_bcast(sh.c[deg], 0);
    double pr = 1, pi = 0;  // start values (1+i)^lane as in cv::solvePoly
    for (int q = 0; q < lane && q < 10; q++) {
        const double nr = pr * 1 - pi * 1, ni = pr * 1 + pi * 1;
        pr = nr;
        pi = ni;
    }
    for (int iter = 0; iter < 300; iter++) {
        double mre = pr, mim = pi;
        bool moving = false;
        double numr = cdeg, numi = 0, denr = cdeg, deni = 0;
#pragma unroll
        for (int j = 0; j < 10; j++) {
            if (j < deg) {  // uniform
                const double rjr = lane_bcast(pr, j), rji = lane_bcast(pi, j);
                // num = num*p + c[deg-j-1]
                const double tr = numr * pr - numi * pi + cf[j];
                const double ti = numr * pi + numi * pr;
                numr = tr;
                numi = ti;
                if (j != lane) {
                    const double dr = pr - rjr, di = pi - rji;
                    if (dr != 0 || di != 0) {
                        const double er = denr * dr - deni * di, ei = denr * di + deni * dr;
                        denr = er;
                        deni = ei;
                    }
                }
            }
        }
        if (lane < deg) {
            const double t = 1. / (denr * denr + deni * deni);
            const double nre = (numr * denr + numi * deni) * t;
            const double nim = (-numr * deni + numi * denr) * t;
            mre = pr - nre;
            mim = pi - nim;
            const double mag = fabs(nre) + fabs(nim);
            moving = mag > 1e-15 * (1.0 + fabs(pr) + fabs(pi));
        }
        pr = mre;
        pi = mim;
        const bool any = __ballot(moving) != 0ull;
#ifdef VO_RANSAC_TIMING
        rt_iters++;
#endif
        if (!any) break;
    }
#ifdef VO_RANSAC_TIMING
    if (lane == 0) atomicAdd(&g_rt[14], rt_iters);
#endif
    RT_MARK(4);

    // ---------------------------------------------------------------- (g) back-substitution, lane per root
    bool ok = false;
    double Ev[9];
    double z1 = 0;
    if (lane < deg) {
        z1 = pr;
        const double zi = pi;
        if (!(fabs(zi) > 1e-10)) {
            const double z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
            double Bt[3][3], W[3], V[3][3];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double* br = sh.b[j];
                // rows of Bt = columns of Bz
                Bt[0][j] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
                Bt[1][j] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
                Bt[2][j] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
            }
            jacobi_rows<3>(Bt, W, V);  // SVD::solveZ: last row of Vt
            if (!(fabs(V[2][2]) < 1e-10)) {
                const double x = V[2][0] / V[2][2], y = V[2][1] / V[2][2];
                double nrm = 0;
#pragma unroll
                for (int e = 0; e < 9; e++) {
                    Ev[e] = sh.Elin[e][0] * x + sh.Elin[e][1] * y + sh.Elin[e][2] * z1 + sh.Elin[e][3];
                    nrm += Ev[e] * Ev[e];
                }
                nrm = sqrt(nrm);
#pragma unroll
                for (int e = 0; e < 9; e++) Ev[e] = Ev[e] / nrm;
                ok = true;
            }
        }
    }
    if (lane < 10) {
        sh.valid[lane] = ok ? 1 : 0;
        sh.zroot[lane] = z1;
    }
    __syncthreads();
    int nm = 0, mypos = 0;
    for (int j = 0; j < 10; j++) {
        const int vj = sh.valid[j];
        nm += vj;
        if (ok && vj && (sh.zroot[j] < z1 || (sh.zroot[j] == z1 && j < lane))) mypos++;
    }
    if (ok) {
#pragma unroll
        for (int e = 0; e < 9; e++) {
            sh.E[mypos][e] = Ev[e];
            models[mypos * 9 + e] = Ev[e];
        }
    }
    if (lane == 0) nmodels_base[(size_t)pair * max_hyp + h] = nm;
    __syncthreads();

    RT_MARK(5);
    // ---------------------------------------------------------------- (h) Sampson scoring, lanes over points
    if (!counts) return;
    // Every point is loaded ONCE and tested against all candidate models (it used to be loaded once per model, and a
    // wavefront waited out the latency of its four loads in every one of n / 64 x nm trips: 0.5 ms for 8192 matches); the
    // loads of the next 64 points are in flight while the current ones are scored.  Counts are wavefront ballots in scalar
    // registers.  Same sampson_err, same comparison: the counts are unchanged.
    if (lane < kMaxModels) sh.cnt[lane] = 0;
    __syncthreads();
    const double thr_lo = (double)thr_sq * (1.0 - 1e-6), thr_hi = (double)thr_sq * (1.0 + 1e-6);
    const double2* __restrict__ q1 = reinterpret_cast<const double2*>(m1);
    const double2* __restrict__ q2 = reinterpret_cast<const double2*>(m2);
    double2 na = make_double2(0, 0), nb = make_double2(0, 0);
    if (lane < n) {
        na = q1[lane];
        nb = q2[lane];
    }
    for (int base = 0; base < n; base += 64) {
        const double2 pa = na, pb = nb;
        const int i = base + lane, inext = i + 64;
        if (inext < n) {
            na = q1[inext];
            nb = q2[inext];
        }
#pragma unroll 1
        for (int m = 0; m < nm; m++) {  // a real loop: ten unrolled copies cost 80 more registers and a wavefront per SIMD
            double E[9];
#pragma unroll
            for (int e = 0; e < 9; e++) E[e] = sh.E[m][e];
            const bool in = i < n && sampson_inlier(E, pa.x, pa.y, pb.x, pb.y, thr_sq, thr_lo, thr_hi);
            const int c = __popcll(__ballot(in));
            if (lane == 0) atomicAdd(&sh.cnt[m], c);  // LDS add without return: nothing waits for it
        }
    }
    __syncthreads();
    if (lane < nm) counts[lane] = sh.cnt[lane];
    RT_MARK(6);
}

// ------------------------------------------------------------------------------------------------
// sequential-semantics scan over hypotheses [scanned, min(h_end, niters))
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_ransac_scan(PairState* __restrict__ st, int max_hyp, int h_end, double prob,
                                                    const int* __restrict__ nmodels_base, const int* __restrict__ counts_base) {
    const int pair = blockIdx.x, lane = threadIdx.x;
    PairState& S = st[pair];
    const int n = S.n_used;
    if (n < 5) return;
    const int* nmodels = nmodels_base + (size_t)pair * max_hyp;
    const int* counts = counts_base + (size_t)pair * max_hyp * kMaxModels;
    int niters = S.niters, best = S.best_count, best_h = S.best_hyp, best_m = S.best_model;
    int hcur = S.scanned, mstart = 0;
    while (hcur < min(h_end, niters)) {
        const int limit = min(h_end, niters);
        const int hh = hcur + lane;
        int found_m = -1, found_c = 0;
        if (hh < limit) {
            const int nm = nmodels[hh];
            const int ms = (lane == 0) ? mstart : 0;
            const int thr = max(best, 4);
            for (int m = ms; m < nm; m++) {
                const int c = counts[hh * kMaxModels + m];
                if (c > thr) {
                    found_m = m;
                    found_c = c;
                    break;
                }
            }
        }
        const unsigned long long msk = __ballot(found_m >= 0);
        if (!msk) {
            hcur = min(hcur + 64, limit);
            mstart = 0;
            continue;
        }
        const int fl = (int)__ffsll((long long)msk) - 1;
        best = __shfl(found_c, fl, 64);
        best_m = __shfl(found_m, fl, 64);
        best_h = hcur + fl;
        niters = d_ransac_update_num_iters(prob, (double)(n - best) / n, 5, niters);
        // The registrator finishes the models of the current hypothesis whatever the new bound says (its loop tests
        // `iter < niters` only between hypotheses): with a high inlier ratio the first good model can pull niters below the
        // current iteration while better models of the same hypothesis are still to come.
        const int nm_h = nmodels[best_h];
        for (int m = best_m + 1; m < nm_h; m++) {
            const int c = counts[best_h * kMaxModels + m];
            if (c > max(best, 4)) {
                best = c;
                best_m = m;
                niters = d_ransac_update_num_iters(prob, (double)(n - best) / n, 5, niters);
            }
        }
        hcur = best_h + 1;
        mstart = 0;
    }
    if (lane == 0) {
        S.niters = niters;
        S.best_count = best;
        S.best_hyp = best_h;
        S.best_model = best_m;
        S.scanned = max(S.scanned, min(h_end, niters));
        // loop counter of the sequential registrator at exit: the bound, or one past the hypothesis whose
        // update pulled the bound below the current iteration
        S.iters_run = best_h >= 0 ? max(min(h_end, niters), min(h_end, best_h + 1)) : min(h_end, niters);
    }
}

// ------------------------------------------------------------------------------------------------
// winner -> E, mask, ordered inlier list, "Few Inliers" gate (camera_utility.hpp:644-661)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_ransac_finish(PairState* __restrict__ st, int max_hyp, int match_cap,
                                                        const double* __restrict__ m1_base, const double* __restrict__ m2_base,
                                                        const float* __restrict__ p1_base, const float* __restrict__ p2_base,
                                                        const double* __restrict__ models_base, const int* __restrict__ nmodels_base,
                                                        uint8_t* __restrict__ mask_base,
                                                        float* __restrict__ in1_base, float* __restrict__ in2_base) {
    __shared__ int s_wtot[16];
    __shared__ int s_base;
    __shared__ double s_E[9];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PairState& S = st[pair];
    const int n = S.n_used;
    const float thr_sq = S.thr_sq;
    uint8_t* mask = mask_base + (size_t)pair * match_cap;
    if (n < 5) {
        for (int i = tid; i < n; i += 1024) mask[i] = 0;
        return;
    }
    int bh = S.best_hyp, bm = S.best_model;
    bool have = bh >= 0;
    if (n == 5) {  // count == modelPoints: first model of the single run, mask all ones
        have = nmodels_base[(size_t)pair * max_hyp] > 0;
        bh = 0;
        bm = 0;
    }
    if (!have) {
        for (int i = tid; i < n; i += 1024) mask[i] = 0;
        if (tid == 0) {
            S.status = VO_ERR_NO_MODEL;
            S.n_inl = 0;
        }
        return;
    }
    if (tid < 9) s_E[tid] = models_base[(((size_t)pair * max_hyp + bh) * kMaxModels + bm) * 9 + tid];
    if (tid == 0) s_base = 0;
    __syncthreads();
    double E[9];
#pragma unroll
    for (int e = 0; e < 9; e++) E[e] = s_E[e];
    const double* m1 = m1_base + (size_t)pair * match_cap * 2;
    const double* m2 = m2_base + (size_t)pair * match_cap * 2;
    const float* p1 = p1_base + (size_t)pair * match_cap * 2;
    const float* p2 = p2_base + (size_t)pair * match_cap * 2;
    float* in1 = in1_base + (size_t)pair * match_cap * 2;
    float* in2 = in2_base + (size_t)pair * match_cap * 2;
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        bool in = false;
        if (i < n) {
            in = (n == 5) ? true : (sampson_err(E, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= thr_sq);
            mask[i] = in ? 1 : 0;
        }
        const unsigned long long m = __ballot(in);
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        if (in) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            in1[2 * pos] = p1[2 * i];
            in1[2 * pos + 1] = p1[2 * i + 1];
            in2[2 * pos] = p2[2 * i];
            in2[2 * pos + 1] = p2[2 * i + 1];
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wtot[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        S.n_inl = s_base;
        for (int e = 0; e < 9; e++) S.E[e] = E[e];
        S.status = VO_OK;
    }
}

// ------------------------------------------------------------------------------------------------
// recoverPose (A.10)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double det3(const double m[3][3]) {
    return m[0][0] * (m[1][1] * m[2][2] - m[1][2] * m[2][1]) - m[0][1] * (m[1][0] * m[2][2] - m[1][2] * m[2][0]) +
           m[0][2] * (m[1][0] * m[2][1] - m[1][1] * m[2][0]);
}

// gate = 1: optRelativePose semantics (skip when the inlier ratio is too low); gate = 0: always
__global__ __launch_bounds__(64) void k_pose_decompose(PairState* __restrict__ st, float inlier_threshold, int gate) {
    const int pair = blockIdx.x;
    if (threadIdx.x != 0) return;
    PairState& S = st[pair];
    S.good[0] = S.good[1] = S.good[2] = S.good[3] = 0;
    S.n_good = 0;
    S.success = 0;
    if (S.status != VO_OK || S.n_inl <= 0) {
        S.n_inl = max(S.n_inl, 0);
        S.niters = -1;  // marks "no pose stage" for the following kernels
        return;
    }
    if (gate && ((float)S.n_inl / (float)S.n_used <= inlier_threshold)) {  // :657 "Few Inliers"
        S.niters = -1;
        return;
    }
    S.niters = 0;
    double At[3][3], W[3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) At[i][k] = S.E[3 * k + i];  // rows of At = columns of E
    jacobi_rows<3>(At, W, V);
    double U[3][3], Vt[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            U[k][i] = At[i][k];
            Vt[i][k] = V[i][k];
        }
    if (det3(U) < 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) U[i][k] = -U[i][k];
    }
    if (det3(Vt) < 0) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) Vt[i][k] = -Vt[i][k];
    }
    const double Wm[3][3] = {{0, 1, 0}, {-1, 0, 0}, {0, 0, 1}}, Wt[3][3] = {{0, -1, 0}, {1, 0, 0}, {0, 0, 1}};
    double UW[3][3];
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const double(*M)[3] = c == 0 ? Wm : Wt;
                UW[i][j] = U[i][0] * M[0][j] + U[i][1] * M[1][j] + U[i][2] * M[2][j];
            }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) S.Rc[c][3 * i + j] = UW[i][0] * Vt[0][j] + UW[i][1] * Vt[1][j] + UW[i][2] * Vt[2][j];
    }
    S.tc[0] = U[0][2];
    S.tc[1] = U[1][2];
    S.tc[2] = U[2][2];
}

__global__ __launch_bounds__(256) void k_pose_cheirality(PairState* __restrict__ st, int match_cap,
                                                         const float* __restrict__ in1_base, const float* __restrict__ in2_base,
                                                         double dist_thresh) {
    // grid = (point blocks, pairs, 4 candidate poses): a thread triangulates its points under ONE candidate (the four DLTs of
    // a point used to run back to back in one thread: 0.29 ms of latency per launch), and a fixed, small number of blocks per
    // pair strides over the inliers instead of one block per 256 slots of capacity (most of which only came to leave)
    const int pair = blockIdx.y, c = blockIdx.z;
    PairState& S = st[pair];
    if (S.niters < 0) return;
    const int n = S.n_inl;
    if (blockIdx.x * 256 >= n) return;
    const int lane = threadIdx.x & 63;
    const double fx = S.K[0], fy = S.K[4], cx = S.K[2], cy = S.K[5];
    const float* in1 = in1_base + (size_t)pair * match_cap * 2;
    const float* in2 = in2_base + (size_t)pair * match_cap * 2;
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    const double* Rc = S.Rc[(c == 0 || c == 2) ? 0 : 1];
    const double sgn = c < 2 ? 1.0 : -1.0;
    double Pk[12];
#pragma unroll
    for (int r = 0; r < 3; r++) {
#pragma unroll
        for (int q = 0; q < 3; q++) Pk[4 * r + q] = Rc[3 * r + q];
        Pk[4 * r + 3] = sgn * S.tc[r];
    }
    int total = 0;
    for (int i0 = blockIdx.x * 256; i0 < n; i0 += gridDim.x * 256) {
        const int i = i0 + threadIdx.x;
        bool good = false;
        if (i < n) {
            const double x1 = ((double)in1[2 * i] - cx) / fx, y1 = ((double)in1[2 * i + 1] - cy) / fy;
            const double x2 = ((double)in2[2 * i] - cx) / fx, y2 = ((double)in2[2 * i + 1] - cy) / fy;
            double Q[4];
            dlt_point(P0, Pk, x1, y1, x2, y2, Q);
            bool m = (Q[2] * Q[3] > 0);
            const double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3], q3 = Q[3] / Q[3];
            m = m && (q2 < dist_thresh);
            const double z = Pk[8] * q0 + Pk[9] * q1 + Pk[10] * q2 + Pk[11] * q3;
            m = m && (z > 0) && (z < dist_thresh);
            good = m;
        }
        total += __popcll(__ballot(good));
    }
    if (lane == 0 && total) atomicAdd(&S.good[c], total);
}

// gate = 1: R/t of the state are overwritten only when VPF >= threshold (optRelativePose);
// gate = 0: plain recoverPose
__global__ __launch_bounds__(64) void k_pose_select(PairState* __restrict__ st, float vpf_threshold, int gate) {
    const int pair = blockIdx.x;
    if (threadIdx.x != 0) return;
    PairState& S = st[pair];
    if (S.niters < 0) return;
    const int g0 = S.good[0], g1 = S.good[1], g2 = S.good[2], g3 = S.good[3];
    int sel;
    if (g0 >= g1 && g0 >= g2 && g0 >= g3)
        sel = 0;
    else if (g1 >= g0 && g1 >= g2 && g1 >= g3)
        sel = 1;
    else if (g2 >= g0 && g2 >= g1 && g2 >= g3)
        sel = 2;
    else
        sel = 3;
    S.n_good = S.good[sel];
    const float vpf = (float)S.n_good / (float)S.n_inl;  // :669
    if (!gate || vpf >= vpf_threshold) {
        const double* Rs = S.Rc[(sel == 0 || sel == 2) ? 0 : 1];
        const double sg = sel < 2 ? 1.0 : -1.0;
        for (int q = 0; q < 9; q++) S.R[q] = Rs[q];
        for (int q = 0; q < 3; q++) S.t[q] = sg * S.tc[q];
        S.success = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// triangulation (A.11, A.12)
// ------------------------------------------------------------------------------------------------
// mode 0: raw triangulatePoints with given P0/P1 (X4 only); mode 1: triangPoints of the reference
__global__ __launch_bounds__(256) void k_triang(PairState* __restrict__ st, int match_cap, const float* __restrict__ in1_base,
                                                const float* __restrict__ in2_base, const double* __restrict__ Pext,
                                                int n_ext, int need_success, float reproj_tol,
                                                float* __restrict__ X4_base, double* __restrict__ Xpre_base,
                                                uint8_t* __restrict__ flag_base) {
    const int pair = blockIdx.y;
    int n;
    double P0[12], P1[12], R[9], t[3], K[9];
    if (Pext) {
        n = n_ext;
#pragma unroll
        for (int q = 0; q < 12; q++) {
            P0[q] = Pext[q];
            P1[q] = Pext[12 + q];
        }
    } else {
        PairState& S = st[pair];
        if (need_success && !S.success) return;
        n = S.n_inl;
#pragma unroll
        for (int q = 0; q < 9; q++) {
            R[q] = S.R[q];
            K[q] = S.K[q];
        }
#pragma unroll
        for (int q = 0; q < 3; q++) t[q] = S.t[q];
        // projectionMatrix(): K * [R|t], K * [I|0]  (camera_utility.hpp:280-286, 316-317)
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                double s0 = 0, s1 = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    s0 += K[3 * i + k] * ((j < 3) ? (k == j ? 1.0 : 0.0) : 0.0);
                    s1 += K[3 * i + k] * ((j < 3) ? R[3 * k + j] : t[k]);
                }
                P0[4 * i + j] = s0;
                P1[4 * i + j] = s1;
            }
    }
    const float* in1 = in1_base + (size_t)pair * match_cap * 2;
    const float* in2 = in2_base + (size_t)pair * match_cap * 2;
    // the blocks of a pair stride over its points (the batch launches a few blocks per pair, not one per 256 slots of capacity)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const float u1 = in1[2 * i], v1 = in1[2 * i + 1], u2 = in2[2 * i], v2 = in2[2 * i + 1];
    double X[4];
    dlt_point(P0, P1, (double)u1, (double)v1, (double)u2, (double)v2, X);
    float h[4];
    float* X4 = X4_base + (size_t)pair * 4 * match_cap;
    const int ld = Pext ? n_ext : match_cap;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        h[q] = (float)X[q];
        X4[(size_t)q * ld + i] = h[q];
    }
    if (Pext) continue;
    // my_convertFromHom (float divide, :544) -> CV_64F (:327)
    const double Xw[3] = {(double)(h[0] / h[3]), (double)(h[1] / h[3]), (double)(h[2] / h[3])};
    const double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    double e[2];
#pragma unroll
    for (int view = 0; view < 2; view++) {
        double x, y, z;
        if (view == 0) {
            x = 1.0 * Xw[0] + 0.0 * Xw[1] + 0.0 * Xw[2] + 0.0;
            y = 0.0 * Xw[0] + 1.0 * Xw[1] + 0.0 * Xw[2] + 0.0;
            z = 0.0 * Xw[0] + 0.0 * Xw[1] + 1.0 * Xw[2] + 0.0;
        } else {
            x = R[0] * Xw[0] + R[1] * Xw[1] + R[2] * Xw[2] + t[0];
            y = R[3] * Xw[0] + R[4] * Xw[1] + R[5] * Xw[2] + t[1];
            z = R[6] * Xw[0] + R[7] * Xw[1] + R[8] * Xw[2] + t[2];
        }
        z = z ? 1. / z : 1;
        x *= z;
        y *= z;
        const double u = x * fx + cx, v = y * fy + cy;
        const double du = (double)(view == 0 ? u1 : u2) - u, dv = (double)(view == 0 ? v1 : v2) - v;
        e[view] = sqrt(du * du + dv * dv);
    }
    const double mean = (e[0] + e[1]) / 2.0;
    const bool keep = (mean < (double)reproj_tol) && (Xw[2] > 0);  // filter_convertWP :559
    flag_base[(size_t)pair * match_cap + i] = keep ? 1 : 0;
    double* Xp = Xpre_base + ((size_t)pair * match_cap + i) * 3;
#pragma unroll
    for (int r = 0; r < 3; r++) Xp[r] = R[3 * r] * Xw[0] + R[3 * r + 1] * Xw[1] + R[3 * r + 2] * Xw[2] + t[r];  // coordTransf
    }
}

// ordered compaction of the kept points, var_avgMat (sequential sums, math_utility.hpp:235-258),
// 3-sigma filter on Z (:575-583); output Xk is [3][match_cap]
__global__ __launch_bounds__(1024) void k_triang_filter(PairState* __restrict__ st, int match_cap,
                                                        const uint8_t* __restrict__ flag_base, const double* __restrict__ Xpre_base,
                                                        double* __restrict__ tmp_base, double* __restrict__ Xk_base) {
    __shared__ int s_wtot[16];
    __shared__ int s_base;
    __shared__ double s_stat[2];
    const int pair = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    PairState& S = st[pair];
    if (!S.success) {
        if (tid == 0) {
            S.n_pre = 0;
            S.n_world = 0;
        }
        return;
    }
    const int n = S.n_inl;
    const uint8_t* flag = flag_base + (size_t)pair * match_cap;
    const double* Xpre = Xpre_base + (size_t)pair * match_cap * 3;
    double* good = tmp_base + (size_t)pair * match_cap * 4;  // [M][3] compacted, then [M] scratch
    double* scr = good + (size_t)match_cap * 3;
    double* Xk = Xk_base + (size_t)pair * 3 * match_cap;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const bool keep = (i < n) && flag[i];
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        if (keep) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            good[3 * pos] = Xpre[3 * i];
            good[3 * pos + 1] = Xpre[3 * i + 1];
            good[3 * pos + 2] = Xpre[3 * i + 2];
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wtot[w];
            s_base += tot;
        }
        __syncthreads();
    }
    const int M = s_base;
    if (M == 0) {
        if (tid == 0) {
            S.n_pre = 0;
            S.n_world = 0;
        }
        return;
    }
    __threadfence_block();
    __syncthreads();
    // mean: the divisions run in parallel, the additions in the reference's order
    for (int i = tid; i < M; i += 1024) scr[i] = good[3 * i + 2] / M;
    __syncthreads();
    if (tid == 0) {
        double mean = 0;
        for (int i = 0; i < M; i++) mean += scr[i];
        s_stat[0] = mean;
    }
    __syncthreads();
    const double mean = s_stat[0];
    for (int i = tid; i < M; i += 1024) {
        const double d = good[3 * i + 2] - mean;
        scr[i] = d * d / M;
    }
    __syncthreads();
    if (tid == 0) {
        double var = 0;
        for (int i = 0; i < M; i++) var += scr[i];
        s_stat[1] = var;
        s_base = 0;
    }
    __syncthreads();
    const double sd3 = 3.0 * sqrt(s_stat[1]);
    const double hi = mean + sd3, lo = mean - sd3;
    for (int base = 0; base < M; base += 1024) {
        const int i = base + tid;
        bool keep = false;
        double z = 0;
        if (i < M) {
            z = good[3 * i + 2];
            keep = (z <= hi) && (z >= lo);
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wtot[wave] = __popcll(m);
        __syncthreads();
        int off = s_base;
        for (int w = 0; w < wave; w++) off += s_wtot[w];
        if (keep) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            Xk[pos] = good[3 * i];
            Xk[(size_t)match_cap + pos] = good[3 * i + 1];
            Xk[(size_t)2 * match_cap + pos] = z;
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += s_wtot[w];
            s_base += tot;
        }
        __syncthreads();
    }
    if (tid == 0) {
        S.n_pre = M;
        S.n_world = s_base;
        S.zstat[0] = mean;
        S.zstat[1] = s_stat[1];
    }
}

// masks + counts of explicit models (inspection entry point vo_sampson_masks)
__global__ __launch_bounds__(256) void k_sampson_masks(const double* __restrict__ m1, const double* __restrict__ m2, int n,
                                                       const double* __restrict__ Es, float thr_sq,
                                                       uint8_t* __restrict__ masks, int* __restrict__ counts) {
    const int model = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    double E[9];
#pragma unroll
    for (int e = 0; e < 9; e++) E[e] = Es[9 * model + e];
    bool in = false;
    if (i < n) {
        in = sampson_err(E, m1[2 * i], m1[2 * i + 1], m2[2 * i], m2[2 * i + 1]) <= thr_sq;
        masks[(size_t)model * n + i] = in ? 1 : 0;
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&counts[model], __popcll(b));
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
constexpr int kRawLen = kMaxHyp * 16 + 4096;  // 32-bit draws kept of a cv::RNG stream
constexpr size_t kMaxRngTables = 8;           // seeds a context keeps resident

static inline uint32_t rng_next(uint64_t& state) {
    state = (uint64_t)(uint32_t)state * 4164903690U + (uint32_t)(state >> 32);
    return (uint32_t)state;
}

// ---- vo_triang_points_noalloc: the one entry point that allocates nothing -------------------------------------------
// Its kernels (k_triang, k_triang_filter) need the same intermediates as vo_triang_points' - staged inlier points, pair state
// (counters, Z statistics), homogeneous points, pre-filter points, filter flags, compaction scratch, output rows - which that
// entry point keeps in hipMalloc'ed pair buffers.  Here they live in this file-scope device array: one copy per GPU, held
// until the process ends, shared by every context on that GPU (overlapping calls on one GPU race; the entry point is
// synchronous, so calls from one thread never overlap).  In host code the array's NAME is only a handle: its device address
// comes from hipGetSymbolAddress, once per context with the context's device current (vo_geometry_init), and that pointer -
// never the name - is what kernels, copies and memsets are given.
constexpr size_t kNaN = VO_NOALLOC_MAX_POINTS;
constexpr size_t na_align(size_t v) { return (v + 255) & ~(size_t)255; }
constexpr size_t kNaSt = 0;
constexpr size_t kNaIn1 = kNaSt + na_align(sizeof(PairState));
constexpr size_t kNaIn2 = kNaIn1 + na_align(kNaN * 2 * sizeof(float));
constexpr size_t kNaX4 = kNaIn2 + na_align(kNaN * 2 * sizeof(float));
constexpr size_t kNaXpre = kNaX4 + na_align(kNaN * 4 * sizeof(float));
constexpr size_t kNaXk = kNaXpre + na_align(kNaN * 3 * sizeof(double));
constexpr size_t kNaFlag = kNaXk + na_align(kNaN * 3 * sizeof(double));
constexpr size_t kNaTmp = kNaFlag + na_align(kNaN);
constexpr size_t kNaBytes = kNaTmp + na_align(kNaN * 4 * sizeof(double));  // 1 852 416 bytes
static __device__ __align__(256) unsigned char scratch[kNaBytes];

int vo_geometry_init(vo_ctx* ctx) {
    int signs[36];
    uint64_t s = 0x12345678;
    for (int i = 0; i < 36; i++) signs[i] = (rng_next(s) & 256) != 0 ? 1 : 0;
    VO_HIP(hipMalloc((void**)&ctx->d_signs, sizeof(signs)));
    VO_HIP(hipMemcpy(ctx->d_signs, signs, sizeof(signs), hipMemcpyHostToDevice));
    // this device's copy of the scratch array (the caller, vo_ctx_create, has made the context's device current)
    unsigned char* ptr = nullptr;
    VO_HIP(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    ctx->d_noalloc = ptr;
    return VO_OK;
}

void vo_geometry_release(vo_ctx* ctx) {
    if (ctx->d_signs) hipFree(ctx->d_signs);
    ctx->d_signs = nullptr;
    for (auto& t : ctx->rng_tables)
        if (t.d_raw) hipFree(t.d_raw);
    ctx->rng_tables.clear();
}

// Device table of the cv::RNG stream for `seed`.  Tables belong to the context and are written exactly once, by a blocking
// copy into memory nothing has read yet - so a batch whose geometry is still in flight on the geometry stream keeps reading
// ITS seed's table while a single call (or another batch) asks for a different seed.  Only when a context has used more
// than kMaxRngTables seeds is the oldest table recycled, after all three streams of the context have drained.
static int rng_table(vo_ctx* ctx, uint64_t seed, const uint32_t** out) {
    for (auto& t : ctx->rng_tables)
        if (t.seed == seed) {
            *out = t.d_raw;
            return VO_OK;
        }
    std::vector<uint32_t> raw(kRawLen);
    uint64_t s = seed ? seed : 0xffffffffULL;  // cv::RNG(0) falls back to 0xffffffff
    for (int i = 0; i < kRawLen; i++) raw[i] = rng_next(s);
    vo_ctx::RngTable t{seed, nullptr};
    if (ctx->rng_tables.size() >= kMaxRngTables) {
        VO_HIP(hipStreamSynchronize(ctx->stream));
        VO_HIP(hipStreamSynchronize(ctx->stream_geo));
        VO_HIP(hipStreamSynchronize(ctx->stream_aux));
        t.d_raw = ctx->rng_tables.front().d_raw;
        ctx->rng_tables.erase(ctx->rng_tables.begin());
    } else {
        VO_HIP(hipMalloc((void**)&t.d_raw, raw.size() * sizeof(uint32_t)));
    }
    if (hipMemcpy(t.d_raw, raw.data(), raw.size() * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess) {
        vo_set_hip_error("hipMemcpy(rng table)", hipGetLastError(), __FILE__, __LINE__);
        hipFree(t.d_raw);
        return VO_ERR_HIP;
    }
    ctx->rng_tables.push_back(t);
    *out = t.d_raw;
    return VO_OK;
}

extern "C" int vo_ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5) {
    if (!idx5 || count < 5 || n_tuples < 0) return VO_ERR_INVALID_ARG;
    uint64_t s = seed ? seed : 0xffffffffULL;
    for (int t = 0; t < n_tuples; t++) {
        int32_t* idx = idx5 + 5 * (size_t)t;
        for (int i = 0; i < 5;) {
            const int v = (int)(rng_next(s) % (unsigned)count);
            bool seen = false;
            for (int q = 0; q < i; q++) seen = seen || idx[q] == v;
            if (!seen) idx[i++] = v;
        }
    }
    return VO_OK;
}

static float thr_sq_of(double thr_px, const double K[9]) {
    const double threshold = thr_px / ((K[0] + K[4]) / 2);
    return (float)(threshold * threshold);
}

// RANSAC (optional) -> pose (optional) -> triangPoints (optional) for every pair of the set.
// The K used for the threshold is the same for all pairs of a set when called from the batch API
// with differing intrinsics the threshold is taken per pair on the device.
int vo_enqueue_geometry(vo_ctx* ctx, PairSet& ps, const GeoOptions& o, hipStream_t stream) {
    const int P = ps.P, mc = ps.match_cap;
    // point blocks per pair of the per-point kernels: a batch strides a few blocks over each pair's points (one block per 256
    // slots of capacity meant 4096 workgroups of 120-144 VGPRs per launch, most of which only came to leave)
    const int nbp = P >= 8 ? std::min((mc + 255) / 256, 8) : (mc + 255) / 256;
    if (o.run_ransac) {
        const uint32_t* d_raw = nullptr;
        VO_TRY(rng_table(ctx, o.seed, &d_raw));
        {
            StageScope sc(ctx, ST_RANSAC_SAMPLE, stream);
            hipLaunchKernelGGL(k_geo_prep, dim3((mc + 255) / 256, P), dim3(256), 0, stream, ps.p1, ps.p2, mc, ps.m1,
                               ps.m2, ps.st, o.max_iters, o.thr_px, 1);
            hipLaunchKernelGGL(k_ransac_tuples, dim3(P), dim3(64), 0, stream, d_raw, kRawLen, ps.st, ps.max_hyp,
                               ps.tuples);
            VO_HIP(hipGetLastError());
        }
        // Hypothesis chunks.  A hypothesis is one wavefront of 136 VGPRs that lives ~0.8 ms whatever the chunk size, so a chunk
        // of 64 x P wavefronts of a batch holds most of the chip's registers for that long and pushes the detection it runs
        // beside off the SIMDs (0.5 ms of a 35 ms step), although the adaptive bound is usually reached within a few
        // hypotheses.  Batches therefore start with 16; a single pair keeps 64 (its chunk is small, and a chunk more is 0.8 ms
        // of latency when the first one does not suffice).
        // Past 256 hypotheses everything that is left goes out as ONE launch: a chunk costs a wavefront's lifetime (~0.1 - 0.3 ms)
        // whatever its size, and wavefronts past the bound the scans have reached by then leave at once.  (If the bound drops
        // inside that chunk the hypotheses behind it were solved for nothing; the in-order scan ignores them.)
        // A few pairs (the synchronous calls: one pair) never fill the chip - 1024 hypotheses are one wavefront per SIMD - so a
        // further chunk boundary buys nothing there and costs a wavefront's lifetime: 64, then everything
        // (configs[4], one pair x 4096 hypotheses x 8192 matches: 2.03 -> 1.54 ms).
        const int bounds[5] = {0, P >= 8 ? 16 : 64, 64, P >= 8 ? 256 : 64, kMaxHyp};
        for (int c = 0; c < 4; c++) {
            const int h0 = bounds[c], h1 = std::min(bounds[c + 1], o.max_iters);
            if (h0 >= h1) continue;  // an empty chunk (a single pair's second bound) or past max_iters
            {
                StageScope sc(ctx, ST_RANSAC_SOLVE_SCORE, stream);
                hipLaunchKernelGGL(k_ransac_hyp, dim3(h1 - h0, P), dim3(64), 0, stream, ps.m1, ps.m2, mc, ps.tuples,
                                   ps.max_hyp, h0, h1, ps.st, ctx->d_signs, ps.models,
                                   ps.nmodels, ps.counts);
                VO_HIP(hipGetLastError());
            }
            {
                StageScope sc(ctx, ST_RANSAC_SELECT, stream);
                hipLaunchKernelGGL(k_ransac_scan, dim3(P), dim3(64), 0, stream, ps.st, ps.max_hyp, h1, o.prob,
                                   ps.nmodels, ps.counts);
                VO_HIP(hipGetLastError());
            }
        }
        {
            StageScope sc(ctx, ST_RANSAC_SELECT, stream);
            hipLaunchKernelGGL(k_ransac_finish, dim3(P), dim3(1024), 0, stream, ps.st, ps.max_hyp, mc, ps.m1, ps.m2,
                               ps.p1, ps.p2, ps.models, ps.nmodels, ps.mask, ps.in1, ps.in2);
            VO_HIP(hipGetLastError());
        }
    }
    if (o.run_pose) {
        StageScope sc(ctx, ST_POSE, stream);
        hipLaunchKernelGGL(k_pose_decompose, dim3(P), dim3(64), 0, stream, ps.st, ctx->prm.inlier_threshold,
                           o.gate ? 1 : 0);
        hipLaunchKernelGGL(k_pose_cheirality, dim3(nbp, P, 4), dim3(256), 0, stream, ps.st, mc, ps.in1,
                           ps.in2, o.dist_thresh);
        hipLaunchKernelGGL(k_pose_select, dim3(P), dim3(64), 0, stream, ps.st, ctx->prm.vpf_threshold, o.gate ? 1 : 0);
        VO_HIP(hipGetLastError());
    }
    if (o.run_triang) {
        StageScope sc(ctx, ST_TRIANGULATE, stream);
        hipLaunchKernelGGL(k_triang, dim3(nbp, P), dim3(256), 0, stream, ps.st, mc, ps.in1, ps.in2,
                           (const double*)nullptr, 0, 1, ctx->prm.reprojection_tolerance, ps.X4, ps.Xpre, ps.flag);
        hipLaunchKernelGGL(k_triang_filter, dim3(P), dim3(1024), 0, stream, ps.st, mc, ps.flag, ps.Xpre, ps.tmp,
                           ps.Xk);
        VO_HIP(hipGetLastError());
    }
    return VO_OK;
}

int vo_enqueue_five_point(vo_ctx* ctx, const double* d_x1, const double* d_x2, int n_hyp, const int* d_tuples,
                          double* d_E, int* d_nm) {
    StageScope sc(ctx, ST_RANSAC_SOLVE_SCORE);
    hipLaunchKernelGGL(k_ransac_hyp, dim3(n_hyp, 1), dim3(64), 0, ctx->stream, d_x1, d_x2, 0, d_tuples, n_hyp, 0, n_hyp,
                       (const PairState*)nullptr, ctx->d_signs, d_E, d_nm, (int*)nullptr);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

int vo_enqueue_sampson_masks(vo_ctx* ctx, const double* d_m1, const double* d_m2, int n, float thr_sq, const double* d_E,
                             int n_models, uint8_t* d_masks, int* d_counts) {
    if (n <= 0 || n_models <= 0) return VO_OK;
    hipLaunchKernelGGL(k_sampson_masks, dim3((n + 255) / 256, n_models), dim3(256), 0, ctx->stream, d_m1, d_m2, n, d_E,
                       thr_sq, d_masks, d_counts);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

int vo_enqueue_triangulate(vo_ctx* ctx, const double* d_P01, const float* d_p1, const float* d_p2, int n, float* d_X4) {
    if (n <= 0) return VO_OK;
    StageScope sc(ctx, ST_TRIANGULATE);
    hipLaunchKernelGGL(k_triang, dim3((n + 255) / 256, 1), dim3(256), 0, ctx->stream, (PairState*)nullptr, 0, d_p1, d_p2,
                       d_P01, n, 0, 0.f, d_X4, (double*)nullptr, (uint8_t*)nullptr);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

float vo_thr_sq(double thr_px, const double K[9]) { return thr_sq_of(thr_px, K); }

int vo_enqueue_normalise(vo_ctx* ctx, PairSet& ps, double thr_px) {
    hipLaunchKernelGGL(k_geo_prep, dim3((ps.match_cap + 255) / 256, ps.P), dim3(256), 0, ctx->stream, ps.p1, ps.p2, ps.match_cap,
                       ps.m1, ps.m2, ps.st, 1, thr_px, 1);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

// ---- a13-a14 without any allocation (no counterpart in the reference; see include/vo_mi355x.h) -----------------------
extern "C" int vo_triang_points_noalloc(vo_ctx* ctx, const float* in1xy, const float* in2xy, int n, const double R[9],
                                        const double t[3], const double K[9], double* Xk, int cap, int* m) {
    if (!ctx || !R || !t || !K || !m || n < 0 || cap < 0 || (n > 0 && (!in1xy || !in2xy))) return VO_ERR_INVALID_ARG;
    *m = 0;
    if (n == 0) return VO_OK;
    if (n > VO_NOALLOC_MAX_POINTS) {
        char msg[160];
        snprintf(msg, sizeof(msg), "vo_triang_points_noalloc: %d points exceed VO_NOALLOC_MAX_POINTS = %d (use vo_triang_points)", n,
                 VO_NOALLOC_MAX_POINTS);
        vo_set_error_text(msg);
        return VO_ERR_CAPACITY;
    }
    if (ctx->profiling) return VO_ERR_UNSUPPORTED;  // the stage timers create events
    const size_t pb = (size_t)n * 2 * sizeof(float), sb = na_align(sizeof(PairState)), wb = (size_t)n * sizeof(double);
    // pinned bounce buffers: sized for the maximum by vo_ctx_create, never grown here
    if (!ctx->d_noalloc || ctx->h_up_bytes < sb + 2 * pb || ctx->h_down_bytes < sb + 3 * wb) return VO_ERR_UNSUPPORTED;
    VO_HIP(hipSetDevice(ctx->device));
    unsigned char* d = ctx->d_noalloc;  // the device address of `scratch` on this GPU
    PairSet v;                          // a view into the scratch array: one pair, VO_NOALLOC_MAX_POINTS matches
    v.P = 1;
    v.kp_cap = 64;
    v.match_cap = (int)kNaN;
    v.max_hyp = 0;
    v.st = reinterpret_cast<PairState*>(d + kNaSt);
    v.in1 = reinterpret_cast<float*>(d + kNaIn1);
    v.in2 = reinterpret_cast<float*>(d + kNaIn2);
    v.X4 = reinterpret_cast<float*>(d + kNaX4);
    v.Xpre = reinterpret_cast<double*>(d + kNaXpre);
    v.Xk = reinterpret_cast<double*>(d + kNaXk);
    v.flag = d + kNaFlag;
    v.tmp = reinterpret_cast<double*>(d + kNaTmp);
    // the whole pair state is written here, so nothing of the previous call survives in it; flags, X4, Xpre, tmp and Xk are
    // written by the kernels for [0, n) before they are read
    PairState* S = reinterpret_cast<PairState*>(ctx->h_up);
    std::memset(S, 0, sizeof(*S));
    S->n_match = S->n_used = S->n_inl = n;
    S->best_hyp = -1;
    S->success = 1;
    std::memcpy(S->K, K, 9 * sizeof(double));
    std::memcpy(S->R, R, 9 * sizeof(double));
    std::memcpy(S->t, t, 3 * sizeof(double));
    std::memcpy(ctx->h_up + sb, in1xy, pb);
    std::memcpy(ctx->h_up + sb + pb, in2xy, pb);
    VO_HIP(hipMemcpyAsync(v.st, ctx->h_up, sizeof(PairState), hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(v.in1, ctx->h_up + sb, pb, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipMemcpyAsync(v.in2, ctx->h_up + sb + pb, pb, hipMemcpyHostToDevice, ctx->stream));
    GeoOptions o;
    o.run_ransac = o.run_pose = false;
    VO_TRY(vo_enqueue_geometry(ctx, v, o, ctx->stream));
    // the number of points kept is known only after the kernels: bring the first n entries of every row down with the state
    VO_HIP(hipMemcpyAsync(ctx->h_down, v.st, sizeof(PairState), hipMemcpyDeviceToHost, ctx->stream));
    for (int r = 0; r < 3; r++)
        VO_HIP(hipMemcpyAsync(ctx->h_down + sb + (size_t)r * wb, v.Xk + (size_t)r * kNaN, wb, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    const PairState* Sd = reinterpret_cast<const PairState*>(ctx->h_down);
    *m = Sd->n_world;
    if (Sd->n_world > cap) return VO_ERR_CAPACITY;
    if (Sd->n_world > 0 && Xk)
        for (int r = 0; r < 3; r++) std::memcpy(Xk + (size_t)r * cap, ctx->h_down + sb + (size_t)r * wb, (size_t)Sd->n_world * sizeof(double));
    return VO_OK;
}
