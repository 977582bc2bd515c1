/*
 * vo_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the OpenCV algorithms the reference front end calls
 * (reference: include/visual_odometry/camera_utility.hpp:742-809, 633-685, 288-351).
 * The arithmetic lives in OpenCV 4.x + opencv_contrib/xfeatures2d (version unpinned by
 * the reference: CMakeLists.txt:21, README.md:9-30), which is NOT present in this
 * repository's build environment.  The restatement follows SURVEY.md Appendix A.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixtures for this path,
 * and OpenCV cannot be run here, so this oracle is pinned only by analytic known-answer
 * tests (tests/test_oracle_*.py), not by outputs of the reference itself.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (libvo_mi355x.so) never links or calls it.
 */
#ifndef VO_ORACLE_H
#define VO_ORACLE_H
#include "../include/vo_mi355x.h"

#ifdef __cplusplus
extern "C" {
#endif

int vo_oracle_set_threads(int n); /* OpenMP threads used by the oracle (0 = all)        */
int vo_oracle_integral(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum);
int vo_oracle_hessian_layer(const uint8_t* img, int rows, int cols, size_t step, int octave, int layer,
                            float* det, float* trace);
int vo_oracle_debug_force_angle(float degrees); /* test hook: see oracle_surf.cpp, g_forced_dir */
int vo_oracle_detect_describe(const vo_params* p, const uint8_t* img, int rows, int cols, size_t step,
                              vo_keypoint* kps, float* desc64, int cap, int* n);
int vo_oracle_match_knn2(const float* descA, int nA, const float* descB, int nB, int32_t* idx2, float* dist2);
/* the same for descriptors of dlen floats (128: SURF's `extended` descriptors) */
int vo_oracle_match_knn2_d(const float* descA, int nA, const float* descB, int nB, int dlen, int32_t* idx2, float* dist2);
int vo_oracle_detect_match(const vo_params* p, const uint8_t* img1, size_t step1, const uint8_t* img2,
                           size_t step2, int rows, int cols, float* p1xy, float* p2xy, vo_dmatch* m, int cap,
                           int* n);
int vo_oracle_ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5);
int vo_oracle_five_point(const double* x1, const double* x2, int n_hyp, double* E_out, int32_t* n_models);
int vo_oracle_sampson_masks(const float* p1xy, const float* p2xy, int n, const double K[9], double thr_px,
                            const double* E, int n_models, uint8_t* masks, int32_t* counts);
int vo_oracle_essential_ransac(const float* p1xy, const float* p2xy, int n, const double K[9], double prob,
                               double thr_px, int max_iters, uint64_t seed, double E[9], uint8_t* mask,
                               int* n_inl, int* iters_run);
int vo_oracle_ransac_replay(const float* p1xy, const float* p2xy, int n, const double K[9], double prob, double thr_px,
                            int max_iters, const double* models, const int32_t* nmodels, double E[9], uint8_t* mask,
                            int* n_inl, int* iters_run);
int vo_oracle_recover_pose(const double E[9], const float* p1xy, const float* p2xy, int n, const double K[9],
                           double dist_thresh, double R[9], double t[3], int* n_good);
int vo_oracle_relative_pose(const vo_params* p, const float* p1xy, const float* p2xy, int n, const double K[9],
                            double R_io[9], double t_io[3], float* in1xy, float* in2xy, int* n_inl,
                            int* success);
int vo_oracle_triangulate(const double P0[12], const double P1[12], const float* p1xy, const float* p2xy,
                          int n, float* X4);
int vo_oracle_triang_points(const vo_params* p, const float* in1xy, const float* in2xy, int n,
                            const double R[9], const double t[3], const double K[9], double* Xk, int cap,
                            int* m);
/* full pair (the unit bench.py's cpu_baseline times): detect x2, match, RANSAC, pose, triangulate */
int vo_oracle_pair(const vo_params* p, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2,
                   int rows, int cols, const double K[9], vo_pair_out* out);

/* N2: get_image (camera_utility.hpp:147-179) on one BGR8 frame.  Every output pointer but gray may be NULL:
 * resized / undist [dst_rows][dst_cols][3], gray0 (before CLAHE) and gray [dst_rows][dst_cols], lut [ty][tx][256],
 * map1 int16 [dst_rows][dst_cols][2], map2 uint16 [dst_rows][dst_cols].                                          */
int vo_oracle_prep(const vo_prep_params* p, const uint8_t* bgr, size_t step, uint8_t* resized, uint8_t* undist,
                   uint8_t* gray0, uint8_t* lut, int16_t* map1, uint16_t* map2, uint8_t* gray);

#ifdef __cplusplus
}
#endif
#endif
