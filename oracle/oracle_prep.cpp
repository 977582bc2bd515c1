// TEST INFRASTRUCTURE ONLY -- see vo_oracle.h.
//
// CPU restatement of the node's preprocessing, get_image (reference include/visual_odometry/camera_utility.hpp:147-179):
//   resize(INTER_AREA) -> undistort(K, [k1 k2 p1 p2]) -> cvtColor(COLOR_RGB2GRAY) on BGR data -> CLAHE(clip 16, 8 x 8).
// PARITY UNPINNED, and more loosely than the SURF path: the arithmetic lives in OpenCV imgproc (version unpinned by the
// reference), SURVEY.md has no appendix for these functions, and they are restated here from the published algorithms:
//   * resize INTER_AREA: ResizeAreaFast (integer scales, (s+2)>>2 for 2x2, cvRound(sum * (1.f/area)) otherwise) and
//     ResizeArea with DecimateAlpha tables (float accumulation in table order) - the same code as the SURF window decimation;
//   * undistort: stripes of (4096 / cols) rows, each with its own inverse of the shifted camera matrix (cv::invert's
//     3 x 3 cofactor formula), distortion model in double with the incremental
//     x += ir[0] walk, maps quantised to 1/32 px (CV_16SC2 + CV_16UC1), remap INTER_LINEAR with the 15-bit fixed-point
//     bilinear table and BORDER_CONSTANT 0;
//   * RGB2GRAY: the 15-bit coefficients of OpenCV >= 4.x (9798, 19235, 3735), first channel weighted 0.299 - on the
//     node's BGR frames that is the BLUE channel;
//   * CLAHE: tile histograms on the image padded bottom/right with BORDER_REFLECT_101 (by tiles - rows % tiles even when
//     only one dimension needs it), clip + redistribution, LUT = cvRound(cdf * 255 / tileArea), bilinear LUT blending.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>

#include "oracle_internal.h"

namespace vo_oracle {

namespace {
struct DecAlpha {
    int si, di;
    float alpha;
};

int area_tab(int ssize, int dsize, double scale, std::vector<DecAlpha>& tab) {
    tab.clear();
    for (int dx = 0; dx < dsize; dx++) {
        double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        double cellWidth = std::min(scale, ssize - fsx1);
        int sx1 = cvCeil(fsx1), sx2 = cvFloor(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tab.push_back({sx1 - 1, dx, (float)((sx1 - fsx1) / cellWidth)});
        for (int sx = sx1; sx < sx2; sx++) tab.push_back({sx, dx, float(1.0 / cellWidth)});
        if (fsx2 - sx2 > 1e-3) tab.push_back({sx2, dx, (float)(std::min(std::min(fsx2 - sx2, 1.), cellWidth) / cellWidth)});
    }
    return (int)tab.size();
}
}  // namespace

// cv::resize(src, dst, Size(dcols, drows), 0, 0, INTER_AREA) for CV_8UC3, shrinking in both directions
void resize_area_bgr(const uint8_t* src, int srows, int scols, size_t sstep, uint8_t* dst, int drows, int dcols) {
    const int cn = 3;
    const double scale_x = (double)scols / dcols, scale_y = (double)srows / drows;
    const int iscale_x = (int)lrint(scale_x), iscale_y = (int)lrint(scale_y);
    const bool fast = std::fabs(scale_x - iscale_x) < DBL_EPSILON && std::fabs(scale_y - iscale_y) < DBL_EPSILON;
    if (fast) {
        const int area = iscale_x * iscale_y;
        const float fscale = 1.f / area;
        for (int dy = 0; dy < drows; dy++)
            for (int dx = 0; dx < dcols; dx++)
                for (int c = 0; c < cn; c++) {
                    int s = 0;
                    for (int sy = 0; sy < iscale_y; sy++)
                        for (int sx = 0; sx < iscale_x; sx++) s += src[(size_t)(dy * iscale_y + sy) * sstep + (dx * iscale_x + sx) * cn + c];
                    uint8_t v;
                    if (area == 1)
                        v = (uint8_t)s;
                    else if (iscale_x == 2 && iscale_y == 2)
                        v = (uint8_t)((s + 2) >> 2);
                    else
                        v = saturate_u8(cvRoundF(s * fscale));
                    dst[((size_t)dy * dcols + dx) * cn + c] = v;
                }
        return;
    }
    std::vector<DecAlpha> xtab, ytab;
    const int xn = area_tab(scols, dcols, scale_x, xtab), yn = area_tab(srows, drows, scale_y, ytab);
    std::vector<float> buf((size_t)dcols * cn), sum((size_t)dcols * cn, 0.f);
    int prev_dy = ytab[0].di;
    for (int j = 0; j < yn; j++) {
        const float beta = ytab[j].alpha;
        const int dy = ytab[j].di;
        const uint8_t* S = src + (size_t)ytab[j].si * sstep;
        std::fill(buf.begin(), buf.end(), 0.f);
        for (int k = 0; k < xn; k++)
            for (int c = 0; c < cn; c++) buf[xtab[k].di * cn + c] += S[xtab[k].si * cn + c] * xtab[k].alpha;
        if (dy != prev_dy) {
            for (int i = 0; i < dcols * cn; i++) {
                dst[(size_t)prev_dy * dcols * cn + i] = saturate_u8(cvRoundF(sum[i]));
                sum[i] = beta * buf[i];
            }
            prev_dy = dy;
        } else {
            for (int i = 0; i < dcols * cn; i++) sum[i] += beta * buf[i];
        }
    }
    for (int i = 0; i < dcols * cn; i++) dst[(size_t)prev_dy * dcols * cn + i] = saturate_u8(cvRoundF(sum[i]));
}

// 32 x 32 bilinear weights in 15-bit fixed point, four per sub-pixel position (x-fraction fastest).
// imgwarp.cpp initInterTab2D: w = saturate_cast<short>(vy * vx * 32768); the products are multiples of 2^-10, so every
// cell sums to 32768 exactly EXCEPT cell (0, 0), whose single weight 32768 saturates to 32767.  OpenCV's sum fix-up looks
// at the "central 2 x 2" of a ksize x ksize kernel starting at ksize / 2 = 1, which for ksize = 2 is entry [3] of the cell
// (and three entries of the next, still zero, cell): the missing 1 is added to w[3].  Cell (0, 0) = {32767, 0, 0, 1}.
void bilinear_tab_i(int16_t* tab /*[1024][4]*/) {
    for (int fy = 0; fy < 32; fy++)
        for (int fx = 0; fx < 32; fx++) {
            const float x = fx * (1.f / 32), y = fy * (1.f / 32);
            const float vx[2] = {1.f - x, x}, vy[2] = {1.f - y, y};
            int16_t* w = tab + (size_t)(fy * 32 + fx) * 4;
            int isum = 0;
            for (int k = 0; k < 4; k++) {
                const float v = vy[k >> 1] * vx[k & 1];
                w[k] = (int16_t)std::min(std::max(cvRoundF(v * 32768.f), -32768), 32767);
                isum += w[k];
            }
            const int diff = isum - 32768;
            if (diff < 0) w[3] = (int16_t)(w[3] - diff);  // diff > 0 cannot happen for the bilinear table
        }
}

// 3 x 3 inverse as cv::invert computes it for n == 3 (cofactors times 1 / det3)
static bool invert3(const double m[9], double t[9]) {
    double d = m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
    if (d == 0.) return false;
    d = 1. / d;
    t[0] = (m[4] * m[8] - m[5] * m[7]) * d;
    t[1] = (m[2] * m[7] - m[1] * m[8]) * d;
    t[2] = (m[1] * m[5] - m[2] * m[4]) * d;
    t[3] = (m[5] * m[6] - m[3] * m[8]) * d;
    t[4] = (m[0] * m[8] - m[2] * m[6]) * d;
    t[5] = (m[2] * m[3] - m[0] * m[5]) * d;
    t[6] = (m[3] * m[7] - m[4] * m[6]) * d;
    t[7] = (m[1] * m[6] - m[0] * m[7]) * d;
    t[8] = (m[0] * m[4] - m[1] * m[3]) * d;
    return true;
}

// cv::undistort's maps for a rows x cols image: map1 = integer source position (x, y), map2 = sub-pixel index fy*32 + fx.
// undistort.cpp: stripes of (4096 / cols) rows; each stripe calls initUndistortRectifyMap(A, dist, I, Ar) with
// Ar(1, 2) = v0 - y0, which inverts Ar and walks x by repeated addition.
void undistort_maps(int rows, int cols, const double K[9], const double dist[4], int16_t* map1, uint16_t* map2) {
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const int stripe0 = std::min(std::max(1, (1 << 12) / std::max(cols, 1)), rows);
    for (int y0 = 0; y0 < rows; y0 += stripe0) {
        const int sh = std::min(stripe0, rows - y0);
        double Ar[9], ir[9];
        std::memcpy(Ar, K, sizeof(Ar));
        Ar[5] = v0 - y0;
        if (!invert3(Ar, ir)) std::memset(ir, 0, sizeof(ir));  // cv::invert zero-fills a singular input's inverse
        for (int i = 0; i < sh; i++) {
            double _x = i * ir[1] + ir[2], _y = i * ir[4] + ir[5], _w = i * ir[7] + ir[8];
            int16_t* m1 = map1 + (size_t)(y0 + i) * cols * 2;
            uint16_t* m2 = map2 + (size_t)(y0 + i) * cols;
            for (int j = 0; j < cols; j++, _x += ir[0], _y += ir[3], _w += ir[6]) {
                const double w = 1. / _w, x = _x * w, y = _y * w;
                const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
                const double kr = (1 + ((0 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
                const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
                const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
                const double u = fx * xd + u0, v = fy * yd + v0;
                // saturate_cast<int>(double) = cvRound with saturation
                const double us = u * 32, vs = v * 32;
                const int iu = us >= 2147483647.0 ? 2147483647 : (us <= -2147483648.0 ? (-2147483647 - 1) : cvRound(us));
                const int iv = vs >= 2147483647.0 ? 2147483647 : (vs <= -2147483648.0 ? (-2147483647 - 1) : cvRound(vs));
                m1[j * 2] = (int16_t)(iu >> 5);
                m1[j * 2 + 1] = (int16_t)(iv >> 5);
                m2[j] = (uint16_t)((iv & 31) * 32 + (iu & 31));
            }
        }
    }
}

// cv::remap(src, dst, map1, map2, INTER_LINEAR, BORDER_CONSTANT, 0) for CV_8UC3
void remap_bilinear_bgr(const uint8_t* src, int rows, int cols, const int16_t* map1, const uint16_t* map2, const int16_t* wtab,
                        uint8_t* dst) {
    const int cn = 3;
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) {
            const int sx = map1[((size_t)y * cols + x) * 2], sy = map1[((size_t)y * cols + x) * 2 + 1];
            const int16_t* w = wtab + (size_t)map2[(size_t)y * cols + x] * 4;
            uint8_t* D = dst + ((size_t)y * cols + x) * cn;
            for (int c = 0; c < cn; c++) {
                auto at = [&](int yy, int xx) -> int {
                    return ((unsigned)xx < (unsigned)cols && (unsigned)yy < (unsigned)rows) ? src[((size_t)yy * cols + xx) * cn + c] : 0;
                };
                const int t = at(sy, sx) * w[0] + at(sy, sx + 1) * w[1] + at(sy + 1, sx) * w[2] + at(sy + 1, sx + 1) * w[3];
                D[c] = saturate_u8((t + (1 << 14)) >> 15);
            }
        }
}

// cvtColor(COLOR_RGB2GRAY) on 3-channel u8: channel 0 takes the "R" weight
void rgb2gray_u8(const uint8_t* src, int rows, int cols, uint8_t* dst) {
    for (size_t i = 0; i < (size_t)rows * cols; i++)
        dst[i] = (uint8_t)((src[3 * i] * 9798 + src[3 * i + 1] * 19235 + src[3 * i + 2] * 3735 + (1 << 14)) >> 15);
}

// cv::createCLAHE(clip, Size(tiles_x, tiles_y))->apply for CV_8UC1.  lut_out (optional): tiles_y * tiles_x * 256 bytes.
void clahe_u8(const uint8_t* src, int rows, int cols, double clip_limit, int tiles_x, int tiles_y, uint8_t* dst, uint8_t* lut_out) {
    int erows = rows, ecols = cols;
    if (!(cols % tiles_x == 0 && rows % tiles_y == 0)) {
        erows = rows + (tiles_y - rows % tiles_y);
        ecols = cols + (tiles_x - cols % tiles_x);
    }
    auto refl = [](int p, int n) {  // BORDER_REFLECT_101
        if (n == 1) return 0;
        while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
        return p;
    };
    const int tw = ecols / tiles_x, th = erows / tiles_y;
    const int tile_total = tw * th;
    const float lut_scale = (float)(256 - 1) / tile_total;
    int clip = 0;
    if (clip_limit > 0.0) {
        clip = (int)(clip_limit * tile_total / 256);
        clip = std::max(clip, 1);
    }
    std::vector<uint8_t> lut((size_t)tiles_x * tiles_y * 256);
    for (int ty = 0; ty < tiles_y; ty++)
        for (int tx = 0; tx < tiles_x; tx++) {
            int hist[256] = {0};
            for (int y = ty * th; y < (ty + 1) * th; y++)
                for (int x = tx * tw; x < (tx + 1) * tw; x++) hist[src[(size_t)refl(y, rows) * cols + refl(x, cols)]]++;
            if (clip > 0) {
                int clipped = 0;
                for (int i = 0; i < 256; i++)
                    if (hist[i] > clip) {
                        clipped += hist[i] - clip;
                        hist[i] = clip;
                    }
                const int batch = clipped / 256;
                int residual = clipped - batch * 256;
                for (int i = 0; i < 256; i++) hist[i] += batch;
                if (residual != 0) {
                    const int step = std::max(256 / residual, 1);
                    for (int i = 0; i < 256 && residual > 0; i += step, residual--) hist[i]++;
                }
            }
            uint8_t* L = lut.data() + (size_t)(ty * tiles_x + tx) * 256;
            int sum = 0;
            for (int i = 0; i < 256; i++) {
                sum += hist[i];
                L[i] = saturate_u8(cvRoundF(sum * lut_scale));
            }
        }
    if (lut_out) std::memcpy(lut_out, lut.data(), lut.size());
    const float inv_tw = 1.0f / tw, inv_th = 1.0f / th;
    for (int y = 0; y < rows; y++) {
        const float tyf = y * inv_th - 0.5f;
        int ty1 = cvFloor(tyf), ty2 = ty1 + 1;
        const float ya = tyf - ty1, ya1 = 1.0f - ya;
        ty1 = std::max(ty1, 0);
        ty2 = std::min(ty2, tiles_y - 1);
        for (int x = 0; x < cols; x++) {
            const float txf = x * inv_tw - 0.5f;
            int tx1 = cvFloor(txf), tx2 = tx1 + 1;
            const float xa = txf - tx1, xa1 = 1.0f - xa;
            tx1 = std::max(tx1, 0);
            tx2 = std::min(tx2, tiles_x - 1);
            const int v = src[(size_t)y * cols + x];
            const uint8_t* P1 = lut.data() + (size_t)ty1 * tiles_x * 256;
            const uint8_t* P2 = lut.data() + (size_t)ty2 * tiles_x * 256;
            const float res = (P1[tx1 * 256 + v] * xa1 + P1[tx2 * 256 + v] * xa) * ya1 + (P2[tx1 * 256 + v] * xa1 + P2[tx2 * 256 + v] * xa) * ya;
            dst[(size_t)y * cols + x] = saturate_u8(cvRoundF(res));
        }
    }
}

}  // namespace vo_oracle
