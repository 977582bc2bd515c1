// TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.  C entry points of the CPU oracle (ctypes-friendly).
#include "vo_oracle.h"

#include <algorithm>
#include <cstring>
#include <vector>

#include "oracle_internal.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace vo_oracle;

static vo_params defaults_or(const vo_params* p) {
    vo_params d;
    std::memset(&d, 0, sizeof(d));
    d.hessian_threshold = 50.f;
    d.n_octaves = 4;
    d.n_octave_layers = 3;
    d.ratio_thresh = 0.7f;
    d.roi_x_lo = 28;
    d.roi_y_lo = 20;
    d.roi_x_hi = 614;
    d.roi_y_hi = 482;
    d.ransac_prob = 0.99;
    d.ransac_threshold = 2.0;
    d.ransac_max_iters = 1000;
    d.ransac_seed = (uint64_t)-1;
    d.inlier_threshold = 0.3f;
    d.vpf_threshold = 0.85f;
    d.distance_threshold = 50.0;
    d.reprojection_tolerance = 0.5f;
    d.kp_capacity = 16384;
    d.match_capacity = 16384;
    return p ? *p : d;
}

extern "C" {

int vo_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n <= 0) n = omp_get_num_procs();
    omp_set_num_threads(n);
    return n;
#else
    (void)n;
    return 1;
#endif
}

int vo_oracle_integral(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum) {
    if (!img || !sum || rows <= 0 || cols <= 0 || step < (size_t)cols) return VO_ERR_INVALID_ARG;
    integral_u8(img, rows, cols, step, sum);
    return VO_OK;
}

int vo_oracle_hessian_layer(const uint8_t* img, int rows, int cols, size_t step, int octave, int layer, float* det,
                            float* trace) {
    if (!img || !det || !trace || rows <= 0 || cols <= 0 || octave < 0 || octave > 3 || layer < 0 || layer > 4)
        return VO_ERR_INVALID_ARG;
    std::vector<int32_t> sum((size_t)(rows + 1) * (cols + 1));
    integral_u8(img, rows, cols, step, sum.data());
    int size, st, lr, lc;
    layer_geometry(rows, cols, octave, layer, &size, &st, &lr, &lc);
    std::memset(det, 0, sizeof(float) * (size_t)lr * lc);
    std::memset(trace, 0, sizeof(float) * (size_t)lr * lc);
    calc_layer_det_trace(sum.data(), rows, cols, size, st, det, trace);
    return VO_OK;
}

int vo_oracle_detect_describe(const vo_params* p, const uint8_t* img, int rows, int cols, size_t step,
                              vo_keypoint* kps, float* desc64, int cap, int* n) {
    if (!img || !kps || !n || rows <= 0 || cols <= 0 || cap < 0) return VO_ERR_INVALID_ARG;
    vo_params P = defaults_or(p);
    std::vector<vo_keypoint> k;
    std::vector<float> d;
    detect_describe(P, img, rows, cols, step, k, desc64 ? &d : nullptr);
    *n = (int)k.size();
    if ((int)k.size() > cap) return VO_ERR_CAPACITY;
    std::copy(k.begin(), k.end(), kps);
    if (desc64) std::copy(d.begin(), d.end(), desc64);
    return VO_OK;
}

// test hook: degrees >= 0 make every rotated descriptor window use that direction; < 0 restores the orientation search
int vo_oracle_debug_force_angle(float degrees) {
    vo_oracle::g_forced_dir = degrees;
    return VO_OK;
}

int vo_oracle_match_knn2(const float* descA, int nA, const float* descB, int nB, int32_t* idx2, float* dist2) {
    if (!descA || !descB || !idx2 || !dist2 || nA < 0 || nB < 0) return VO_ERR_INVALID_ARG;
    match_knn2(descA, nA, descB, nB, idx2, dist2);
    return VO_OK;
}

int vo_oracle_match_knn2_d(const float* descA, int nA, const float* descB, int nB, int dlen, int32_t* idx2, float* dist2) {
    if (!descA || !descB || !idx2 || !dist2 || nA < 0 || nB < 0 || dlen < 1) return VO_ERR_INVALID_ARG;
    match_knn2(descA, nA, descB, nB, idx2, dist2, dlen);
    return VO_OK;
}

int vo_oracle_detect_match(const vo_params* p, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2,
                           int rows, int cols, float* p1xy, float* p2xy, vo_dmatch* m, int cap, int* n) {
    if (!img1 || !img2 || !n) return VO_ERR_INVALID_ARG;
    vo_params P = defaults_or(p);
    std::vector<float> a, b;
    std::vector<vo_dmatch> mm;
    int cnt = detect_match(P, img1, step1, img2, step2, rows, cols, a, b, mm, nullptr, nullptr);
    *n = cnt;
    if (cnt > cap) return VO_ERR_CAPACITY;
    if (p1xy) std::copy(a.begin(), a.end(), p1xy);
    if (p2xy) std::copy(b.begin(), b.end(), p2xy);
    if (m) std::copy(mm.begin(), mm.end(), m);
    return VO_OK;
}

int vo_oracle_ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5) {
    if (!idx5 || count < 5 || n_tuples < 0) return VO_ERR_INVALID_ARG;
    ransac_subsets(seed, count, n_tuples, idx5);
    return VO_OK;
}

int vo_oracle_five_point(const double* x1, const double* x2, int n_hyp, double* E_out, int32_t* n_models) {
    if (!x1 || !x2 || !E_out || !n_models || n_hyp < 0) return VO_ERR_INVALID_ARG;
#pragma omp parallel for schedule(dynamic, 4)
    for (int h = 0; h < n_hyp; h++) {
        double E[90];
        std::memset(E, 0, sizeof(E));
        n_models[h] = five_point(x1 + 10 * (size_t)h, x2 + 10 * (size_t)h, E);
        std::memcpy(E_out + 90 * (size_t)h, E, sizeof(E));
    }
    return VO_OK;
}

int vo_oracle_sampson_masks(const float* p1xy, const float* p2xy, int n, const double K[9], double thr_px,
                            const double* E, int n_models, uint8_t* masks, int32_t* counts) {
    if (!p1xy || !p2xy || !K || !E || !masks || !counts || n < 0) return VO_ERR_INVALID_ARG;
    double fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    std::vector<double> m1(2 * (size_t)n), m2(2 * (size_t)n);
    for (int i = 0; i < n; i++) {
        m1[2 * i] = ((double)p1xy[2 * i] - cx) / fx;
        m1[2 * i + 1] = ((double)p1xy[2 * i + 1] - cy) / fy;
        m2[2 * i] = ((double)p2xy[2 * i] - cx) / fx;
        m2[2 * i + 1] = ((double)p2xy[2 * i + 1] - cy) / fy;
    }
    double threshold = thr_px / ((fx + fy) / 2);
    float t = (float)(threshold * threshold);
    std::vector<float> err(n);
    for (int k = 0; k < n_models; k++) {
        sampson_errors(m1.data(), m2.data(), n, E + 9 * (size_t)k, err.data());
        int c = 0;
        for (int i = 0; i < n; i++) c += (masks[(size_t)k * n + i] = (uint8_t)(err[i] <= t));
        counts[k] = c;
    }
    return VO_OK;
}

int vo_oracle_essential_ransac(const float* p1xy, const float* p2xy, int n, const double K[9], double prob,
                               double thr_px, int max_iters, uint64_t seed, double E[9], uint8_t* mask, int* n_inl,
                               int* iters_run) {
    if (!p1xy || !p2xy || !K || !E || !mask || n < 0) return VO_ERR_INVALID_ARG;
    return essential_ransac(p1xy, p2xy, n, K, prob, thr_px, max_iters, seed, E, mask, n_inl, iters_run);
}

// RANSACPointSetRegistrator::run with externally supplied candidate models per hypothesis
// (models [max_iters][10][9], nmodels [max_iters]); used to check the GPU's scoring + sequential scan.
int vo_oracle_ransac_replay(const float* p1xy, const float* p2xy, int n, const double K[9], double prob, double thr_px,
                            int max_iters, const double* models, const int32_t* nmodels, double E[9], uint8_t* mask,
                            int* n_inl, int* iters_run) {
    if (!p1xy || !p2xy || !K || !E || !mask || !models || !nmodels || n <= 5) return VO_ERR_INVALID_ARG;
    return essential_ransac(p1xy, p2xy, n, K, prob, thr_px, max_iters, 1, E, mask, n_inl, iters_run, models, nmodels);
}

int vo_oracle_recover_pose(const double E[9], const float* p1xy, const float* p2xy, int n, const double K[9],
                           double dist_thresh, double R[9], double t[3], int* n_good) {
    if (!E || !p1xy || !p2xy || !K || !R || !t || n < 0) return VO_ERR_INVALID_ARG;
    return recover_pose(E, p1xy, p2xy, n, K, dist_thresh, R, t, n_good);
}

// optRelativePose, ESSENTIAL branch (camera_utility.hpp:639-685)
int vo_oracle_relative_pose(const vo_params* p, const float* p1xy, const float* p2xy, int n, const double K[9],
                            double R_io[9], double t_io[3], float* in1xy, float* in2xy, int* n_inl, int* success) {
    if (!p1xy || !p2xy || !K || !R_io || !t_io || !n_inl || !success) return VO_ERR_INVALID_ARG;
    vo_params P = defaults_or(p);
    *success = 0;
    *n_inl = 0;
    std::vector<uint8_t> mask(std::max(n, 1), 0);
    double E[9];
    int inl = 0;
    int st = essential_ransac(p1xy, p2xy, n, K, P.ransac_prob, P.ransac_threshold, P.ransac_max_iters, P.ransac_seed,
                              E, mask.data(), &inl, nullptr);
    if (st != VO_OK) return st;  // OpenCV would hand an empty E to recoverPose and throw
    // extract_Inlier (:203-245)
    std::vector<float> a, b;
    for (int i = 0; i < n; i++)
        if (mask[i]) {
            a.push_back(p1xy[2 * i]);
            a.push_back(p1xy[2 * i + 1]);
            b.push_back(p2xy[2 * i]);
            b.push_back(p2xy[2 * i + 1]);
        }
    *n_inl = inl;
    if (in1xy) std::copy(a.begin(), a.end(), in1xy);
    if (in2xy) std::copy(b.begin(), b.end(), in2xy);
    if ((float)inl / (float)n <= P.inlier_threshold) return VO_OK;  // "Few Inliers" (:657)
    double R[9], t[3];
    int valid = 0;
    recover_pose(E, a.data(), b.data(), inl, K, P.distance_threshold, R, t, &valid);
    float vpf = (float)valid / (float)inl;
    if (vpf >= P.vpf_threshold) {
        std::memcpy(R_io, R, sizeof(R));
        std::memcpy(t_io, t, sizeof(t));
        *success = 1;
    }
    return VO_OK;
}

int vo_oracle_triangulate(const double P0[12], const double P1[12], const float* p1xy, const float* p2xy, int n,
                          float* X4) {
    if (!P0 || !P1 || !p1xy || !p2xy || !X4 || n < 0) return VO_ERR_INVALID_ARG;
    std::vector<double> x1(n), y1(n), x2(n), y2(n), X(4 * (size_t)n);
    for (int i = 0; i < n; i++) {
        x1[i] = p1xy[2 * i];
        y1[i] = p1xy[2 * i + 1];
        x2[i] = p2xy[2 * i];
        y2[i] = p2xy[2 * i + 1];
    }
    triangulate(P0, P1, x1.data(), y1.data(), x2.data(), y2.data(), n, X.data());
    for (size_t i = 0; i < X.size(); i++) X4[i] = (float)X[i];
    return VO_OK;
}

int vo_oracle_triang_points(const vo_params* p, const float* in1xy, const float* in2xy, int n, const double R[9],
                            const double t[3], const double K[9], double* Xk, int cap, int* m) {
    if (!in1xy || !in2xy || !R || !t || !K || !m || n < 0) return VO_ERR_INVALID_ARG;
    vo_params P = defaults_or(p);
    std::vector<double> X;
    int M = triang_points(P, in1xy, in2xy, n, R, t, K, X);
    *m = M;
    if (M > cap) return VO_ERR_CAPACITY;
    for (int j = 0; j < M; j++)
        for (int r = 0; r < 3; r++) Xk[(size_t)r * cap + j] = X[3 * (size_t)j + r];
    return VO_OK;
}

int vo_oracle_pair(const vo_params* p, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2, int rows,
                   int cols, const double K[9], vo_pair_out* out) {
    if (!img1 || !img2 || !K || !out) return VO_ERR_INVALID_ARG;
    vo_params P = defaults_or(p);
    std::memset(out, 0, sizeof(*out));
    std::vector<float> a, b;
    std::vector<vo_dmatch> mm;
    int n = detect_match(P, img1, step1, img2, step2, rows, cols, a, b, mm, &out->n_kp1, &out->n_kp2);
    out->n_match = n;
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(out->R, I3, sizeof(I3));
    std::vector<uint8_t> mask(std::max(n, 1), 0);
    int st = essential_ransac(a.data(), b.data(), n, K, P.ransac_prob, P.ransac_threshold, P.ransac_max_iters,
                              P.ransac_seed, out->E, mask.data(), &out->n_inl, &out->iters);
    out->status = st;
    if (st != VO_OK) return VO_OK;
    std::vector<float> ia, ib;
    for (int i = 0; i < n; i++)
        if (mask[i]) {
            ia.push_back(a[2 * i]);
            ia.push_back(a[2 * i + 1]);
            ib.push_back(b[2 * i]);
            ib.push_back(b[2 * i + 1]);
        }
    if ((float)out->n_inl / (float)n <= P.inlier_threshold) return VO_OK;
    double R[9], t[3];
    recover_pose(out->E, ia.data(), ib.data(), out->n_inl, K, P.distance_threshold, R, t, &out->n_good);
    if ((float)out->n_good / (float)out->n_inl >= P.vpf_threshold) {
        std::memcpy(out->R, R, sizeof(R));
        std::memcpy(out->t, t, sizeof(t));
        out->success = 1;
        std::vector<double> X;
        out->n_world = triang_points(P, ia.data(), ib.data(), out->n_inl, R, t, K, X);
    }
    return VO_OK;
}

int vo_oracle_prep(const vo_prep_params* p, const uint8_t* bgr, size_t step, uint8_t* resized, uint8_t* undist,
                   uint8_t* gray0, uint8_t* lut, int16_t* map1, uint16_t* map2, uint8_t* gray) {
    if (!p || !bgr || !gray) return VO_ERR_INVALID_ARG;
    const int R = p->dst_rows, C = p->dst_cols;
    if (R > p->src_rows || C > p->src_cols || R < 1 || C < 1) return VO_ERR_UNSUPPORTED;
    const size_t px = (size_t)R * C;
    std::vector<uint8_t> rs(px * 3), ud(px * 3), g0(px);
    if (R != p->src_rows || C != p->src_cols)
        resize_area_bgr(bgr, p->src_rows, p->src_cols, step, rs.data(), R, C);
    else
        for (int r = 0; r < R; r++) std::memcpy(rs.data() + (size_t)r * C * 3, bgr + (size_t)r * step, (size_t)C * 3);
    std::vector<int16_t> m1(px * 2), wtab(1024 * 4);
    std::vector<uint16_t> m2(px);
    bilinear_tab_i(wtab.data());
    undistort_maps(R, C, p->K, p->dist, m1.data(), m2.data());
    remap_bilinear_bgr(rs.data(), R, C, m1.data(), m2.data(), wtab.data(), ud.data());
    rgb2gray_u8(ud.data(), R, C, g0.data());
    clahe_u8(g0.data(), R, C, p->clip_limit, p->tiles_x, p->tiles_y, gray, lut);
    if (resized) std::memcpy(resized, rs.data(), rs.size());
    if (undist) std::memcpy(undist, ud.data(), ud.size());
    if (gray0) std::memcpy(gray0, g0.data(), g0.size());
    if (map1) std::memcpy(map1, m1.data(), m1.size() * sizeof(int16_t));
    if (map2) std::memcpy(map2, m2.data(), m2.size() * sizeof(uint16_t));
    return VO_OK;
}

}  // extern "C"
