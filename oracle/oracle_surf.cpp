// TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.  PARITY UNPINNED (no reference fixtures exist).
//
// CPU restatement of cv::xfeatures2d::SURF_Impl::detectAndCompute as configured by the
// reference (SURF::create(50,4,3,false,false), camera_utility.hpp:746-750) and of the
// brute-force 2-NN matcher + ratio/ROI filter (camera_utility.hpp:759-808), following
// SURVEY.md Appendix A.1-A.7.  OpenCV (opencv_contrib xfeatures2d/src/surf.cpp,
// imgproc/src/resize.cpp, core mathfuncs) is not vendored by the reference; the section
// letters below refer to that appendix.
//
// Built with -ffp-contract=off: every float/double operation rounds once, in the order written.
#include "oracle_internal.h"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

namespace vo_oracle {

// ---------------------------------------------------------------- A.1 step 2: integral
void integral_u8(const uint8_t* img, int rows, int cols, size_t step, int32_t* sum) {
    const int sw = cols + 1;
    for (int x = 0; x < sw; x++) sum[x] = 0;
    for (int y = 0; y < rows; y++) {
        const uint8_t* src = img + (size_t)y * step;
        int32_t* cur = sum + (size_t)(y + 1) * sw;
        const int32_t* prev = sum + (size_t)y * sw;
        cur[0] = 0;
        int32_t run = 0;
        for (int x = 0; x < cols; x++) {
            run += src[x];
            cur[x + 1] = prev[x + 1] + run;
        }
    }
}

// ---------------------------------------------------------------- A.2 box patterns
struct SurfHF {
    int p0, p1, p2, p3;
    float w;
};

static void resizeHaarPattern(const int src[][5], SurfHF* dst, int n, int oldSize, int newSize, int widthStep) {
    float ratio = (float)newSize / oldSize;
    for (int k = 0; k < n; k++) {
        int dx1 = cvRoundF(ratio * src[k][0]);
        int dy1 = cvRoundF(ratio * src[k][1]);
        int dx2 = cvRoundF(ratio * src[k][2]);
        int dy2 = cvRoundF(ratio * src[k][3]);
        dst[k].p0 = dy1 * widthStep + dx1;
        dst[k].p1 = dy2 * widthStep + dx1;
        dst[k].p2 = dy1 * widthStep + dx2;
        dst[k].p3 = dy2 * widthStep + dx2;
        dst[k].w = src[k][4] / ((float)(dx2 - dx1) * (dy2 - dy1));
    }
}

static inline float calcHaarPattern(const int32_t* origin, const SurfHF* f, int n) {
    double d = 0;
    for (int k = 0; k < n; k++) {
        // int box sum -> float for the multiply, product accumulated in double (A.2)
        int32_t box = (int32_t)((uint32_t)origin[f[k].p0] + (uint32_t)origin[f[k].p3] -
                                (uint32_t)origin[f[k].p1] - (uint32_t)origin[f[k].p2]);
        d += (float)box * f[k].w;
    }
    return (float)d;
}

static const int kDxS[3][5] = {{0, 2, 3, 7, 1}, {3, 2, 6, 7, -2}, {6, 2, 9, 7, 1}};
static const int kDyS[3][5] = {{2, 0, 7, 3, 1}, {2, 3, 7, 6, -2}, {2, 6, 7, 9, 1}};
static const int kDxyS[4][5] = {{1, 1, 4, 4, 1}, {5, 1, 8, 4, -1}, {1, 5, 4, 8, -1}, {5, 5, 8, 8, 1}};

void layer_geometry(int rows, int cols, int octave, int layer, int* size, int* step, int* lrows, int* lcols) {
    *size = (9 + 6 * layer) << octave;
    *step = 1 << octave;
    *lrows = rows / *step;
    *lcols = cols / *step;
}

// calcLayerDetAndTrace (A.2).  det/trace are lrows*lcols, pre-zeroed by the caller.
void calc_layer_det_trace(const int32_t* sum, int rows, int cols, int size, int sampleStep, float* det,
                          float* trace) {
    const int sw = cols + 1;
    if (size > rows || size > cols) return;
    SurfHF Dx[3], Dy[3], Dxy[4];
    resizeHaarPattern(kDxS, Dx, 3, 9, size, sw);
    resizeHaarPattern(kDyS, Dy, 3, 9, size, sw);
    resizeHaarPattern(kDxyS, Dxy, 4, 9, size, sw);
    const int samples_i = 1 + (rows - size) / sampleStep;
    const int samples_j = 1 + (cols - size) / sampleStep;
    const int margin = (size / 2) / sampleStep;
    const int lcols = cols / sampleStep;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < samples_i; i++) {
        const int32_t* sum_ptr = sum + (size_t)(i * sampleStep) * sw;
        float* det_ptr = det + (size_t)(i + margin) * lcols + margin;
        float* trace_ptr = trace + (size_t)(i + margin) * lcols + margin;
        for (int j = 0; j < samples_j; j++) {
            float dx = calcHaarPattern(sum_ptr, Dx, 3);
            float dy = calcHaarPattern(sum_ptr, Dy, 3);
            float dxy = calcHaarPattern(sum_ptr, Dxy, 4);
            sum_ptr += sampleStep;
            det_ptr[j] = dx * dy - 0.81f * dxy * dxy;
            trace_ptr[j] = dx + dy;
        }
    }
}

// ---------------------------------------------------------------- A.3 interpolateKeypoint
static int interpolateKeypoint(const float N9[3][9], int dx, int dy, int ds, vo_keypoint& kpt) {
    float b0 = -(N9[1][5] - N9[1][3]) / 2;
    float b1 = -(N9[1][7] - N9[1][1]) / 2;
    float b2 = -(N9[2][4] - N9[0][4]) / 2;
    float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
    float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
    float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
    float a10 = a01;
    float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
    float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
    float a20 = a02;
    float a21 = a12;
    float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
    // Matx33f::solve(DECOMP_LU) == Cramer's rule in float (A.3)
    float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
    float x0 = 0, x1 = 0, x2 = 0;
    if (d != 0) {
        d = 1 / d;
        x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
        x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
        x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
    }
    bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && std::fabs(x0) <= 1 && std::fabs(x1) <= 1 && std::fabs(x2) <= 1;
    if (ok) {
        kpt.x += x0 * dx;
        kpt.y += x1 * dy;
        kpt.size = (float)cvRoundF(kpt.size + x2 * ds);
    }
    return ok;
}

// KeypointGreater (A.4): response desc, size desc, octave desc, pt.y desc, pt.x asc
static bool keypointGreater(const vo_keypoint& a, const vo_keypoint& b) {
    if (a.response > b.response) return true;
    if (a.response < b.response) return false;
    if (a.size > b.size) return true;
    if (a.size < b.size) return false;
    if (a.octave > b.octave) return true;
    if (a.octave < b.octave) return false;
    if (a.y > b.y) return true;
    if (a.y < b.y) return false;
    return a.x < b.x;
}

// fastHessianDetector (A.2-A.4).  nOctaves / nOctaveLayers are the reference's 4 / 3 (camera_utility.hpp:35-36) unless the
// caller asks otherwise (the octaves are independent of each other: fewer octaves = the same keypoints without those of the
// dropped octaves; the layer loops below are A.2's, written for any layer count)
void fast_hessian(const int32_t* sum, int rows, int cols, float hessianThreshold, std::vector<vo_keypoint>& kps, int nOctaves,
                  int nOctaveLayers) {
    const int nTotal = (nOctaveLayers + 2) * nOctaves;
    std::vector<std::vector<float>> dets(nTotal), traces(nTotal);
    std::vector<int> sizes(nTotal), steps(nTotal);
    int idx = 0;
    for (int o = 0; o < nOctaves; o++)
        for (int l = 0; l < nOctaveLayers + 2; l++, idx++) {
            int size, step, lr, lc;
            layer_geometry(rows, cols, o, l, &size, &step, &lr, &lc);
            sizes[idx] = size;
            steps[idx] = step;
            dets[idx].assign((size_t)lr * lc, 0.f);
            traces[idx].assign((size_t)lr * lc, 0.f);
            calc_layer_det_trace(sum, rows, cols, size, step, dets[idx].data(), traces[idx].data());
        }
    kps.clear();
    std::mutex mtx;
    for (int o = 0; o < nOctaves; o++)
        for (int l = 1; l <= nOctaveLayers; l++) {
            const int layer = o * (nOctaveLayers + 2) + l;
            const int size = sizes[layer];
            const int sampleStep = steps[layer];
            const int layer_rows = rows / sampleStep;
            const int layer_cols = cols / sampleStep;
            const int margin = (sizes[layer + 1] / 2) / sampleStep + 1;
            const int step = layer_cols;
            const float* D0 = dets[layer - 1].data();
            const float* D1 = dets[layer].data();
            const float* D2 = dets[layer + 1].data();
            const float* T1 = traces[layer].data();
#pragma omp parallel for schedule(dynamic, 8)
            for (int i = margin; i < layer_rows - margin; i++) {
                for (int j = margin; j < layer_cols - margin; j++) {
                    float val0 = D1[(size_t)i * step + j];
                    if (!(val0 > hessianThreshold)) continue;
                    int sum_i = sampleStep * (i - (size / 2) / sampleStep);
                    int sum_j = sampleStep * (j - (size / 2) / sampleStep);
                    const float* det1 = D0 + (size_t)i * step + j;
                    const float* det2 = D1 + (size_t)i * step + j;
                    const float* det3 = D2 + (size_t)i * step + j;
                    float N9[3][9] = {
                        {det1[-step - 1], det1[-step], det1[-step + 1], det1[-1], det1[0], det1[1], det1[step - 1],
                         det1[step], det1[step + 1]},
                        {det2[-step - 1], det2[-step], det2[-step + 1], det2[-1], det2[0], det2[1], det2[step - 1],
                         det2[step], det2[step + 1]},
                        {det3[-step - 1], det3[-step], det3[-step + 1], det3[-1], det3[0], det3[1], det3[step - 1],
                         det3[step], det3[step + 1]}};
                    bool is_max = true;
                    for (int a = 0; a < 3 && is_max; a++)
                        for (int b = 0; b < 9; b++) {
                            if (a == 1 && b == 4) continue;
                            if (!(val0 > N9[a][b])) {
                                is_max = false;
                                break;
                            }
                        }
                    if (!is_max) continue;
                    float center_i = sum_i + (size - 1) * 0.5f;
                    float center_j = sum_j + (size - 1) * 0.5f;
                    vo_keypoint kpt;
                    kpt.x = center_j;
                    kpt.y = center_i;
                    kpt.size = (float)sizes[layer];
                    kpt.angle = -1;
                    kpt.response = val0;
                    kpt.octave = o;
                    float tr = T1[(size_t)i * step + j];
                    kpt.class_id = (tr > 0) - (tr < 0);
                    int ds = size - sizes[layer - 1];
                    if (interpolateKeypoint(N9, sampleStep, sampleStep, ds, kpt)) {
                        std::lock_guard<std::mutex> g(mtx);
                        kps.push_back(kpt);
                    }
                }
            }
        }
    std::sort(kps.begin(), kps.end(), keypointGreater);
}

// ---------------------------------------------------------------- A.5 helpers
// cv::getGaussianKernel(n, sigma, CV_32F) for the non-fixed-table case (A.5)
void gaussian_kernel_f32(int n, double sigma, float* cf) {
    double scale2X = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = std::exp(scale2X * x * x);
        cf[i] = (float)t;
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

// cv::fastAtan2 (degrees, [0,360)) -- the polynomial shared by cv::phase (A.5)
float fastAtan2(float y, float x) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = std::fabs(x), ay = std::fabs(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ---------------------------------------------------------------- A.6.1 resize INTER_AREA (u8, 1 channel)
struct DecimateAlpha {
    int si, di;
    float alpha;
};

static int computeResizeAreaTab(int ssize, int dsize, double scale, DecimateAlpha* tab) {
    int k = 0;
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale;
        double fsx2 = fsx1 + scale;
        double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = cvCeil(fsx1), sx2 = cvFloor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx1 - 1;
            tab[k++].alpha = (float)((sx1 - fsx1) / cellWidth);
        }
        for (int sx = sx1; sx < sx2; sx++) {
            tab[k].di = dx;
            tab[k].si = sx;
            tab[k++].alpha = float(1.0 / cellWidth);
        }
        if (fsx2 - sx2 > 1e-3) {
            tab[k].di = dx;
            tab[k].si = sx2;
            tab[k++].alpha = (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth);
        }
    }
    return k;
}

// win: ssize x ssize u8 (ssize >= dsize), dst: dsize x dsize u8
void resize_area_u8(const uint8_t* win, int ssize, uint8_t* dst, int dsize) {
    double scale = (double)ssize / dsize;
    int iscale = (int)lrint(scale);  // saturate_cast<int>(double) == cvRound
    bool is_area_fast = std::fabs(scale - iscale) < DBL_EPSILON;
    if (is_area_fast) {
        if (iscale == 1) {
            for (int i = 0; i < dsize * dsize; i++) dst[i] = win[i];
            return;
        }
        if (iscale == 2) {
            for (int dy = 0; dy < dsize; dy++)
                for (int dx = 0; dx < dsize; dx++) {
                    const uint8_t* S = win + (size_t)(2 * dy) * ssize + 2 * dx;
                    dst[dy * dsize + dx] = (uint8_t)((S[0] + S[1] + S[ssize] + S[ssize + 1] + 2) >> 2);
                }
            return;
        }
        int area = iscale * iscale;
        float fscale = 1.f / area;
        for (int dy = 0; dy < dsize; dy++)
            for (int dx = 0; dx < dsize; dx++) {
                int s = 0;
                for (int sy = 0; sy < iscale; sy++)
                    for (int sx = 0; sx < iscale; sx++) s += win[(size_t)(dy * iscale + sy) * ssize + dx * iscale + sx];
                dst[dy * dsize + dx] = saturate_u8(cvRoundF(s * fscale));
            }
        return;
    }
    std::vector<DecimateAlpha> xtab(ssize * 2 + 2), ytab(ssize * 2 + 2);
    int xtab_size = computeResizeAreaTab(ssize, dsize, scale, xtab.data());
    int ytab_size = computeResizeAreaTab(ssize, dsize, scale, ytab.data());
    std::vector<float> buf(dsize), sum(dsize);
    int prev_dy = ytab[0].di;
    for (int dx = 0; dx < dsize; dx++) sum[dx] = 0;
    for (int j = 0; j < ytab_size; j++) {
        float beta = ytab[j].alpha;
        int dy = ytab[j].di, sy = ytab[j].si;
        const uint8_t* S = win + (size_t)sy * ssize;
        for (int dx = 0; dx < dsize; dx++) buf[dx] = 0;
        for (int k = 0; k < xtab_size; k++) buf[xtab[k].di] += S[xtab[k].si] * xtab[k].alpha;
        if (dy != prev_dy) {
            for (int dx = 0; dx < dsize; dx++) {
                dst[prev_dy * dsize + dx] = saturate_u8(cvRoundF(sum[dx]));
                sum[dx] = beta * buf[dx];
            }
            prev_dy = dy;
        } else {
            for (int dx = 0; dx < dsize; dx++) sum[dx] += beta * buf[dx];
        }
    }
    for (int dx = 0; dx < dsize; dx++) dst[prev_dy * dsize + dx] = saturate_u8(cvRoundF(sum[dx]));
}

// ---------------------------------------------------------------- A.5 + A.6: SURFInvoker
static const int ORI_RADIUS = 6, ORI_WIN = 60, PATCH_SZ = 20;
static const int SURF_ORI_SEARCH_INC = 5;
static const float SURF_ORI_SIGMA = 2.5f, SURF_DESC_SIGMA = 3.3f;

struct InvokerTables {
    int nOriSamples;
    int aptx[169], apty[169];
    float aptw[169];
    float DW[PATCH_SZ * PATCH_SZ];
    InvokerTables() {
        float G_ori[2 * ORI_RADIUS + 1];
        gaussian_kernel_f32(2 * ORI_RADIUS + 1, SURF_ORI_SIGMA, G_ori);
        nOriSamples = 0;
        for (int i = -ORI_RADIUS; i <= ORI_RADIUS; i++)
            for (int j = -ORI_RADIUS; j <= ORI_RADIUS; j++)
                if (i * i + j * j <= ORI_RADIUS * ORI_RADIUS) {
                    aptx[nOriSamples] = i;
                    apty[nOriSamples] = j;
                    aptw[nOriSamples++] = G_ori[i + ORI_RADIUS] * G_ori[j + ORI_RADIUS];
                }
        float G_desc[PATCH_SZ];
        gaussian_kernel_f32(PATCH_SZ, SURF_DESC_SIGMA, G_desc);
        for (int i = 0; i < PATCH_SZ; i++)
            for (int j = 0; j < PATCH_SZ; j++) DW[i * PATCH_SZ + j] = G_desc[i] * G_desc[j];
    }
};

// One keypoint of SURFInvoker::operator().  Returns false if the keypoint is deleted.
// test hook (vo_oracle_debug_force_angle): when >= 0 the rotated-window path uses this direction instead of the one it found, so
// that the axis-aligned `upright` branch below can be checked against the rotated branch at 270 degrees
float g_forced_dir = -1.f;

// upright (SURF's `upright` flag, false in the reference, camera_utility.hpp:38): no orientation search - the angle is 270 and the
// window is an axis-aligned pixel rectangle.  That branch is NOT part of SURVEY Appendix A; it restates the published OpenCV source
// (xfeatures2d surf.cpp, SURFInvoker) and nothing in the reference exercises it: parity unpinned like the rest, one step further.
static bool describe_one(const InvokerTables& T, const uint8_t* img, size_t imgstep, int rows, int cols,
                         const int32_t* sum, vo_keypoint& kp, float* vec /*64 (128: extended) or null*/, std::vector<uint8_t>& winbuf, bool upright,
                         bool extended) {
    const int sw = cols + 1;
    const int sum_rows = rows + 1, sum_cols = cols + 1;
    static const int dx_s[2][5] = {{0, 0, 2, 4, -1}, {2, 0, 4, 4, 1}};
    static const int dy_s[2][5] = {{0, 0, 4, 2, 1}, {0, 2, 4, 4, -1}};
    float X[169], Y[169], angle[169];
    float size = kp.size;
    float cx = kp.x, cy = kp.y;
    float s = size * 1.2f / 9.0f;
    int grad_wav_size = 2 * cvRoundF(2 * s);
    if (sum_rows < grad_wav_size || sum_cols < grad_wav_size) return false;
    float descriptor_dir = 360.f - 90.f;
    if (!upright) {
        SurfHF dx_t[2], dy_t[2];
        resizeHaarPattern(dx_s, dx_t, 2, 4, grad_wav_size, sw);
        resizeHaarPattern(dy_s, dy_t, 2, 4, grad_wav_size, sw);
        int nangle = 0;
        for (int kk = 0; kk < T.nOriSamples; kk++) {
            int x = cvRoundF(cx + T.aptx[kk] * s - (float)(grad_wav_size - 1) / 2);
            int y = cvRoundF(cy + T.apty[kk] * s - (float)(grad_wav_size - 1) / 2);
            if (y < 0 || y >= sum_rows - grad_wav_size || x < 0 || x >= sum_cols - grad_wav_size) continue;
            const int32_t* ptr = sum + (size_t)y * sw + x;
            float vx = calcHaarPattern(ptr, dx_t, 2);
            float vy = calcHaarPattern(ptr, dy_t, 2);
            X[nangle] = vx * T.aptw[kk];
            Y[nangle] = vy * T.aptw[kk];
            nangle++;
        }
        if (nangle == 0) return false;
        for (int j = 0; j < nangle; j++) angle[j] = fastAtan2(Y[j], X[j]);  // cv::phase(X, Y, angle, true)
        float bestx = 0, besty = 0, descriptor_mod = 0;
        for (int i = 0; i < 360; i += SURF_ORI_SEARCH_INC) {
            float sumx = 0, sumy = 0, temp_mod;
            for (int j = 0; j < nangle; j++) {
                int d = std::abs(cvRoundF(angle[j]) - i);
                if (d < ORI_WIN / 2 || d > 360 - ORI_WIN / 2) {
                    sumx += X[j];
                    sumy += Y[j];
                }
            }
            temp_mod = sumx * sumx + sumy * sumy;
            if (temp_mod > descriptor_mod) {
                descriptor_mod = temp_mod;
                bestx = sumx;
                besty = sumy;
            }
        }
        descriptor_dir = fastAtan2(-besty, bestx);
        if (g_forced_dir >= 0.f) descriptor_dir = g_forced_dir;
    }
    kp.angle = descriptor_dir;
    if (!vec) return true;

    // ---- A.6 descriptor
    int win_size = (int)((PATCH_SZ + 1) * s);
    if ((int)winbuf.size() < win_size * win_size) winbuf.resize((size_t)win_size * win_size);
    uint8_t* WIN = winbuf.data();
    if (!upright) {
        descriptor_dir *= (float)(3.1415926535897932384626433832795 / 180);
        // std::sin/std::cos(float) of the reference's libm; the oracle DEFINES them as the
        // double-precision function rounded to float (correctly rounded but for double-rounding
        // corner cases), so that the GPU can reproduce the value bit-for-bit.
        float sin_dir = -(float)std::sin((double)descriptor_dir);
        float cos_dir = (float)std::cos((double)descriptor_dir);
        float win_offset = -(float)(win_size - 1) / 2;
        float start_x = cx + win_offset * cos_dir + win_offset * sin_dir;
        float start_y = cy - win_offset * sin_dir + win_offset * cos_dir;
        int ncols1 = cols - 1, nrows1 = rows - 1;
        for (int i = 0; i < win_size; i++, start_x += sin_dir, start_y += cos_dir) {
            double pixel_x = start_x;
            double pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_x += cos_dir, pixel_y -= sin_dir) {
                int ix = cvFloor(pixel_x), iy = cvFloor(pixel_y);
                if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
                    float a = (float)(pixel_x - ix), b = (float)(pixel_y - iy);
                    const uint8_t* p = img + (size_t)iy * imgstep + ix;
                    WIN[i * win_size + j] = (uint8_t)cvRoundF(p[0] * (1.f - a) * (1.f - b) + p[1] * a * (1.f - b) +
                                                              p[imgstep] * (1.f - a) * b + p[imgstep + 1] * a * b);
                } else {
                    int x = std::min(std::max(cvRound(pixel_x), 0), ncols1);
                    int y = std::min(std::max(cvRound(pixel_y), 0), nrows1);
                    WIN[i * win_size + j] = img[(size_t)y * imgstep + x];
                }
            }
        }
    } else {
        // descriptor_dir == 90 degrees, sin_dir == 1, cos_dir == 0: an extract-rect with clamped borders
        float win_offset = -(float)(win_size - 1) / 2;
        int start_x = cvRoundF(cx + win_offset);
        int start_y = cvRoundF(cy - win_offset);
        for (int i = 0; i < win_size; i++, start_x++) {
            int pixel_x = start_x, pixel_y = start_y;
            for (int j = 0; j < win_size; j++, pixel_y--) {
                int x = std::min(std::max(pixel_x, 0), cols - 1);
                int y = std::min(std::max(pixel_y, 0), rows - 1);
                WIN[i * win_size + j] = img[(size_t)y * imgstep + x];
            }
        }
    }
    uint8_t PATCH[PATCH_SZ + 1][PATCH_SZ + 1];
    resize_area_u8(WIN, win_size, &PATCH[0][0], PATCH_SZ + 1);
    float DX[PATCH_SZ][PATCH_SZ], DY[PATCH_SZ][PATCH_SZ];
    for (int i = 0; i < PATCH_SZ; i++)
        for (int j = 0; j < PATCH_SZ; j++) {
            float dw = T.DW[i * PATCH_SZ + j];
            float vx = (PATCH[i][j + 1] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i + 1][j]) * dw;
            float vy = (PATCH[i + 1][j] - PATCH[i][j] + PATCH[i + 1][j + 1] - PATCH[i][j + 1]) * dw;
            DX[i][j] = vx;
            DY[i][j] = vy;
        }
    const int dlen = extended ? 128 : 64;
    for (int kk = 0; kk < dlen; kk++) vec[kk] = 0;
    double square_mag = 0;
    float* v = vec;
    if (extended) {
        // SURF's `extended` flag (false in the reference, camera_utility.hpp:37): the 128-bin descriptor.  NOT part of SURVEY
        // Appendix A; restates the published OpenCV source (xfeatures2d surf.cpp, SURFInvoker) - parity unpinned, one step further.
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                for (int y = i * 5; y < i * 5 + 5; y++)
                    for (int x = j * 5; x < j * 5 + 5; x++) {
                        float tx = DX[y][x], ty = DY[y][x];
                        if (ty >= 0) {
                            v[0] += tx;
                            v[1] += (float)std::fabs(tx);
                        } else {
                            v[2] += tx;
                            v[3] += (float)std::fabs(tx);
                        }
                        if (tx >= 0) {
                            v[4] += ty;
                            v[5] += (float)std::fabs(ty);
                        } else {
                            v[6] += ty;
                            v[7] += (float)std::fabs(ty);
                        }
                    }
                for (int kk = 0; kk < 8; kk++) square_mag += v[kk] * v[kk];
                v += 8;
            }
    } else
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            for (int y = i * 5; y < i * 5 + 5; y++)
                for (int x = j * 5; x < j * 5 + 5; x++) {
                    float tx = DX[y][x], ty = DY[y][x];
                    v[0] += tx;
                    v[1] += ty;
                    v[2] += (float)std::fabs(tx);
                    v[3] += (float)std::fabs(ty);
                }
            for (int kk = 0; kk < 4; kk++) square_mag += v[kk] * v[kk];
            v += 4;
        }
    float scale = (float)(1. / (std::sqrt(square_mag) + FLT_EPSILON));
    for (int kk = 0; kk < dlen; kk++) vec[kk] *= scale;
    return true;
}

// detectAndCompute (A.1)
void detect_describe(const vo_params& P, const uint8_t* img, int rows, int cols, size_t step,
                     std::vector<vo_keypoint>& kps, std::vector<float>* desc) {
    std::vector<int32_t> sum((size_t)(rows + 1) * (cols + 1));
    integral_u8(img, rows, cols, step, sum.data());
    fast_hessian(sum.data(), rows, cols, P.hessian_threshold, kps, P.n_octaves, P.n_octave_layers);
    if (!desc) return;
    static const InvokerTables T;
    const int N = (int)kps.size();
    const int dlen = P.extended ? 128 : 64;
    std::vector<float> d((size_t)N * dlen);
    std::vector<uint8_t> keep(N, 0);
#pragma omp parallel
    {
        std::vector<uint8_t> winbuf;
#pragma omp for schedule(dynamic, 16)
        for (int k = 0; k < N; k++)
            keep[k] = describe_one(T, img, step, rows, cols, sum.data(), kps[k], d.data() + (size_t)k * dlen, winbuf, P.upright != 0,
                                   P.extended != 0);
    }
    // step 5: drop deleted keypoints preserving order
    int w = 0;
    for (int k = 0; k < N; k++)
        if (keep[k]) {
            if (w != k) {
                kps[w] = kps[k];
                std::memcpy(d.data() + (size_t)w * dlen, d.data() + (size_t)k * dlen, dlen * sizeof(float));
            }
            w++;
        }
    kps.resize(w);
    d.resize((size_t)w * dlen);
    desc->swap(d);
}

// ---------------------------------------------------------------- A.7 brute-force 2-NN
void match_knn2(const float* A, int nA, const float* B, int nB, int32_t* idx2, float* dist2, int dlen) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < nA; i++) {
        const float* a = A + (size_t)i * dlen;
        float d0 = FLT_MAX, d1 = FLT_MAX;
        int i0 = -1, i1 = -1;
        for (int j = 0; j < nB; j++) {
            const float* b = B + (size_t)j * dlen;
            float s = 0;
            for (int k = 0; k < dlen; k++) {
                float t = a[k] - b[k];
                s += t * t;
            }
            // insertion with strict '<' : ties keep the lower train index (A.7)
            if (s < d0) {
                d1 = d0;
                i1 = i0;
                d0 = s;
                i0 = j;
            } else if (s < d1) {
                d1 = s;
                i1 = j;
            }
        }
        idx2[2 * i] = i0;
        idx2[2 * i + 1] = i1;
        dist2[2 * i] = i0 >= 0 ? std::sqrt(d0) : FLT_MAX;
        dist2[2 * i + 1] = i1 >= 0 ? std::sqrt(d1) : FLT_MAX;
    }
}

// opt_DetectFeatures (camera_utility.hpp:742-809) with exact BF in place of FLANN
int detect_match(const vo_params& P, const uint8_t* img1, size_t step1, const uint8_t* img2, size_t step2, int rows,
                 int cols, std::vector<float>& p1, std::vector<float>& p2, std::vector<vo_dmatch>& m, int* nk1,
                 int* nk2) {
    std::vector<vo_keypoint> k1, k2;
    std::vector<float> d1, d2;
    detect_describe(P, img1, rows, cols, step1, k1, &d1);
    detect_describe(P, img2, rows, cols, step2, k2, &d2);
    if (nk1) *nk1 = (int)k1.size();
    if (nk2) *nk2 = (int)k2.size();
    p1.clear();
    p2.clear();
    m.clear();
    if (k1.empty() || k2.size() < 2) return 0;  // reference: UB when the train set has < 2 rows (:776)
    std::vector<int32_t> idx(2 * k1.size());
    std::vector<float> dist(2 * k1.size());
    match_knn2(d1.data(), (int)k1.size(), d2.data(), (int)k2.size(), idx.data(), dist.data(), P.extended ? 128 : 64);
    for (size_t i = 0; i < k1.size(); i++) {
        bool lowe = dist[2 * i] < P.ratio_thresh * dist[2 * i + 1];
        bool roi = (k1[i].x >= P.roi_x_lo) && (k1[i].x <= P.roi_x_hi) && (k1[i].y >= P.roi_y_lo) &&
                   (k1[i].y <= P.roi_y_hi);
        if (lowe && roi) {
            vo_dmatch dm;
            dm.queryIdx = (int)i;
            dm.trainIdx = idx[2 * i];
            dm.imgIdx = 0;
            dm.distance = dist[2 * i];
            m.push_back(dm);
            p1.push_back(k1[i].x);
            p1.push_back(k1[i].y);
            p2.push_back(k2[dm.trainIdx].x);
            p2.push_back(k2[dm.trainIdx].y);
        }
    }
    return (int)m.size();
}

}  // namespace vo_oracle
