// TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
#ifndef VO_ORACLE_INTERNAL_H
#define VO_ORACLE_INTERNAL_H
#include <cmath>
#include <cstdint>
#include <vector>

#include "vo_oracle.h"

namespace vo_oracle {

// OpenCV rounding helpers (SURVEY Appendix A preamble): cvRound = round-half-to-even
static inline int cvRound(double v) { return (int)lrint(v); }
static inline int cvRoundF(float v) { return (int)lrintf(v); }
static inline int cvFloor(double v) {
    int i = (int)v;
    return i - (i > v);
}
static inline int cvCeil(double v) {
    int i = (int)v;
    return i + (i < v);
}
static inline uint8_t saturate_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// oracle_surf.cpp
void integral_u8(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum);
void layer_geometry(int rows, int cols, int octave, int layer, int* size, int* step, int* lrows, int* lcols);
void calc_layer_det_trace(const int32_t* sum, int rows, int cols, int size, int sampleStep, float* det, float* trace);
extern float g_forced_dir;  // test hook, oracle_surf.cpp
void fast_hessian(const int32_t* sum, int rows, int cols, float hessianThreshold, std::vector<vo_keypoint>& kps, int nOctaves = 4,
                  int nOctaveLayers = 3);
void gaussian_kernel_f32(int n, double sigma, float* cf);
float fastAtan2(float y, float x);
void resize_area_u8(const uint8_t* win, int ssize, uint8_t* dst, int dsize);
void detect_describe(const vo_params& P, const uint8_t* img, int rows, int cols, size_t step,
                     std::vector<vo_keypoint>& kps, std::vector<float>* desc);
void match_knn2(const float* A, int nA, const float* B, int nB, int32_t* idx2, float* dist2, int dlen = 64);
int detect_match(const vo_params& P, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2, int rows,
                 int cols, std::vector<float>& p1, std::vector<float>& p2, std::vector<vo_dmatch>& m, int* nk1,
                 int* nk2);

// oracle_geometry.cpp
struct RNG {  // cv::RNG multiply-with-carry (A.8)
    uint64_t state;
    explicit RNG(uint64_t s) : state(s ? s : 0xffffffffULL) {}
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};
void jacobi_svd(double* At, int astep, double* W, double* Vt, int vstep, int m, int n, int n1);
void ransac_subsets(uint64_t seed, int count, int n_tuples, int32_t* idx5);
int five_point(const double x1[10], const double x2[10], double E_out[90]);
void sampson_errors(const double* m1, const double* m2, int n, const double E[9], float* err);
int ransac_update_num_iters(double p, double ep, int modelPoints, int maxIters);
int essential_ransac(const float* p1, const float* p2, int n, const double K[9], double prob, double thr_px,
                     int max_iters, uint64_t seed, double E[9], uint8_t* mask, int* n_inl, int* iters_run,
                     const double* replay_models = nullptr, const int* replay_nmodels = nullptr);
int recover_pose(const double E[9], const float* p1, const float* p2, int n, const double K[9], double dist_thresh,
                 double R[9], double t[3], int* n_good);
void triangulate(const double P0[12], const double P1[12], const double* x1, const double* y1, const double* x2,
                 const double* y2, int n, double* X4 /*[4][n]*/);
int triang_points(const vo_params& P, const float* in1, const float* in2, int n, const double R[9], const double t[3],
                  const double K[9], std::vector<double>& Xk /*[m][3]*/);


// oracle_prep.cpp (N2: get_image, camera_utility.hpp:147-179)
void resize_area_bgr(const uint8_t* src, int srows, int scols, size_t sstep, uint8_t* dst, int drows, int dcols);
void bilinear_tab_i(int16_t* tab /*[1024][4]*/);
void undistort_maps(int rows, int cols, const double K[9], const double dist[4], int16_t* map1, uint16_t* map2);
void remap_bilinear_bgr(const uint8_t* src, int rows, int cols, const int16_t* map1, const uint16_t* map2, const int16_t* wtab,
                        uint8_t* dst);
void rgb2gray_u8(const uint8_t* src, int rows, int cols, uint8_t* dst);
void clahe_u8(const uint8_t* src, int rows, int cols, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst, uint8_t* lut_out);

}  // namespace vo_oracle
#endif
