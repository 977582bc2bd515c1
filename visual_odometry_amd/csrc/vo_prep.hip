// N2: the node's per-frame preprocessing, get_image (reference include/visual_odometry/camera_utility.hpp:147-179),
// on the device: INTER_AREA resize (:119-136) -> undistort (:155) -> RGB2GRAY on BGR data (:160) -> CLAHE (:165-169).
//
// Every kernel here is a streaming pass over a 2560 x 2056 x 3 source (15.8 MB) and a 640 x 505 working frame: HBM-bound
// by construction, a few microseconds of traffic each.  Integer / fixed-point arithmetic is OpenCV's (15-bit bilinear
// weights, 15-bit gray coefficients, integer histograms); the float steps (area weights, LUT scale, LUT blending) keep
// OpenCV's operation order with contraction off, so the result is bit-exact against oracle/oracle_prep.cpp.
//
// The undistortion maps depend on the camera only: they are built once per vo_prep on the host, in double, following
// cv::undistort's striped initUndistortRectifyMap walk (x advances by repeated addition of 1/fx along a row).
#include <algorithm>
#include <cfloat>
#include <cmath>

#include "vo_common.hpp"

namespace {

struct AreaTap {
    int si;
    float alpha;
};

inline int ceil_i(double v) {
    int i = (int)v;
    return i + (i < v);
}
inline int floor_i(double v) {
    int i = (int)v;
    return i - (i > v);
}

// DecimateAlpha table of one axis, grouped by destination index: taps of dst d are tap[ofs[d] .. ofs[d + 1])
void area_taps(int ssize, int dsize, double scale, std::vector<AreaTap>& tap, std::vector<int>& ofs) {
    tap.clear();
    ofs.assign(dsize + 1, 0);
    for (int dx = 0; dx < dsize; dx++) {
        ofs[dx] = (int)tap.size();
        const double fsx1 = dx * scale, fsx2 = fsx1 + scale;
        const double cell = std::min(scale, ssize - fsx1);
        int sx1 = ceil_i(fsx1), sx2 = floor_i(fsx2);
        sx2 = std::min(sx2, ssize - 1);
        sx1 = std::min(sx1, sx2);
        if (sx1 - fsx1 > 1e-3) tap.push_back({sx1 - 1, (float)((sx1 - fsx1) / cell)});
        for (int sx = sx1; sx < sx2; sx++) tap.push_back({sx, float(1.0 / cell)});
        if (fsx2 - sx2 > 1e-3) tap.push_back({sx2, (float)(std::min(std::min(fsx2 - sx2, 1.), cell) / cell)});
    }
    ofs[dsize] = (int)tap.size();
}

__device__ __forceinline__ uint8_t sat_u8(int v) { return (uint8_t)min(max(v, 0), 255); }

// ResizeArea (fractional scale): float accumulation in table order, rows outer.  One lane per destination BYTE (pixel,
// channel): neighbouring lanes read neighbouring source bytes.  This is the any-scale form; its loops are serial chains
// of dependent loads, so the scales the node uses go through k_resize_area_u below.
__global__ __launch_bounds__(256) void k_resize_area(const uint8_t* __restrict__ src, int sstep, const AreaTap* __restrict__ xt,
                                                      const int* __restrict__ xo, const AreaTap* __restrict__ yt,
                                                      const int* __restrict__ yo, uint8_t* __restrict__ dst, int drows, int dcols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (e >= dcols * 3) return;
    const int dx = e / 3, c = e - dx * 3;
    const int k0 = xo[dx], k1 = xo[dx + 1];
    float s = 0.f;
    for (int j = yo[dy]; j < yo[dy + 1]; j++) {
        const float beta = yt[j].alpha;
        const uint8_t* S = src + (size_t)yt[j].si * sstep + c;
        float b = 0.f;
        for (int k = k0; k < k1; k++) b += S[xt[k].si * 3] * xt[k].alpha;
        s += beta * b;  // 0.f + x == x: the first row needs no special case
    }
    dst[(size_t)dy * dcols * 3 + e] = sat_u8((int)rintf(s));
}

// The same sums with at most MX x MY taps per destination byte (the taps of one destination index are consecutive source
// indices).  Absent taps carry weight 0 and re-read the last valid byte: x + 0.f * v == x exactly for these non-negative
// sums, so the loops unroll completely and all MX * MY byte loads are in flight at once.
template <int MX, int MY>
__global__ __launch_bounds__(256) void k_resize_area_u(const uint8_t* __restrict__ src, int sstep, const AreaTap* __restrict__ xt,
                                                        const int* __restrict__ xo, const AreaTap* __restrict__ yt,
                                                        const int* __restrict__ yo, uint8_t* __restrict__ dst, int drows, int dcols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (e >= dcols * 3) return;
    const int dx = e / 3, c = e - dx * 3;
    const int k0 = xo[dx], nx = xo[dx + 1] - k0, j0 = yo[dy], ny = yo[dy + 1] - j0;
    const int sx0 = xt[k0].si, sy0 = yt[j0].si;
    float ax[MX], ay[MY];
#pragma unroll
    for (int k = 0; k < MX; k++) ax[k] = k < nx ? xt[k0 + k].alpha : 0.f;
#pragma unroll
    for (int j = 0; j < MY; j++) ay[j] = j < ny ? yt[j0 + j].alpha : 0.f;
    const uint8_t* S = src + (size_t)sy0 * sstep + sx0 * 3 + c;
    uint8_t v[MY][MX];
#pragma unroll
    for (int j = 0; j < MY; j++)
#pragma unroll
        for (int k = 0; k < MX; k++) v[j][k] = S[(size_t)min(j, ny - 1) * sstep + min(k, nx - 1) * 3];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MY; j++) {
        float b = 0.f;
#pragma unroll
        for (int k = 0; k < MX; k++) b += v[j][k] * ax[k];
        s += ay[j] * b;
    }
    dst[(size_t)dy * dcols * 3 + e] = sat_u8((int)rintf(s));
}

// k_resize_area_u with the source bytes staged through LDS: one block = 256 consecutive destination bytes of one
// destination row; the (at most MY) source rows it needs are read once, with aligned 16-byte loads, into LDS and the
// taps are LDS byte reads.  Needs 16-byte aligned source rows (sstep % 16 == 0): the node's 2560-pixel rows are.
constexpr int kResizePitch = 2048 + 64;  // bytes of one staged source row: 86 pixels * 7 columns * 3 + alignment slack
template <int MX, int MY>
__global__ __launch_bounds__(256) void k_resize_area_lds(const uint8_t* __restrict__ src, int sstep, const AreaTap* __restrict__ xt,
                                                          const int* __restrict__ xo, const AreaTap* __restrict__ yt,
                                                          const int* __restrict__ yo, uint8_t* __restrict__ dst, int drows, int dcols) {
    __shared__ __attribute__((aligned(16))) uint8_t s_row[MY][kResizePitch];
    const int t = threadIdx.x, e0 = blockIdx.x * 256, dy = blockIdx.y;
    const int e_last = min(e0 + 255, dcols * 3 - 1);
    const int j0 = yo[dy], ny = yo[dy + 1] - j0, sy0 = yt[j0].si;
    const int b_lo = (xt[xo[e0 / 3]].si * 3) & ~15;                        // first staged byte of a source row (16-aligned)
    const int b_hi = (xt[xo[e_last / 3 + 1] - 1].si * 3 + 3 + 15) & ~15;   // one past the last, rounded up
    const int chunks = (b_hi - b_lo) >> 4;
    for (int i = t; i < chunks * ny; i += 256) {
        const int j = i / chunks, ch = i - j * chunks;
        *(uint4*)&s_row[j][ch * 16] = *(const uint4*)(src + (size_t)(sy0 + j) * sstep + b_lo + ch * 16);
    }
    __syncthreads();
    const int e = e0 + t;
    if (e >= dcols * 3) return;
    const int dx = e / 3, c = e - dx * 3;
    const int k0 = xo[dx], nx = xo[dx + 1] - k0;
    const int off = xt[k0].si * 3 + c - b_lo;
    float ax[MX], ay[MY];
#pragma unroll
    for (int k = 0; k < MX; k++) ax[k] = k < nx ? xt[k0 + k].alpha : 0.f;
#pragma unroll
    for (int j = 0; j < MY; j++) ay[j] = j < ny ? yt[j0 + j].alpha : 0.f;
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MY; j++) {
        float b = 0.f;
#pragma unroll
        for (int k = 0; k < MX; k++) b += s_row[min(j, ny - 1)][off + min(k, nx - 1) * 3] * ax[k];
        s += ay[j] * b;
    }
    dst[(size_t)dy * dcols * 3 + e] = sat_u8((int)rintf(s));
}

// ResizeAreaFast (integer scale): exact box sums, one lane per destination byte.  IS > 0: square IS x IS boxes unrolled.
template <int IS>
__global__ __launch_bounds__(256) void k_resize_area_int(const uint8_t* __restrict__ src, int sstep, int isx, int isy,
                                                          uint8_t* __restrict__ dst, int drows, int dcols) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (e >= dcols * 3) return;
    const int dx = e / 3, c = e - dx * 3;
    if (IS > 0) isx = isy = IS;
    int s = 0;
    const uint8_t* p = src + (size_t)dy * isy * sstep + (size_t)dx * isx * 3 + c;
    if (IS > 0) {
#pragma unroll
        for (int y = 0; y < IS; y++)
#pragma unroll
            for (int x = 0; x < IS; x++) s += p[(size_t)y * sstep + x * 3];
    } else {
        for (int y = 0; y < isy; y++)
            for (int x = 0; x < isx; x++) s += p[(size_t)y * sstep + x * 3];
    }
    const int area = isx * isy;
    uint8_t v;
    if (area == 1)
        v = (uint8_t)s;
    else if (isx == 2 && isy == 2)
        v = (uint8_t)((s + 2) >> 2);
    else
        v = sat_u8((int)rintf(s * (1.f / area)));
    dst[(size_t)dy * dcols * 3 + e] = v;
}

// remap(INTER_LINEAR, BORDER_CONSTANT 0) with the fixed-point maps, then RGB2GRAY of the rounded BGR bytes.
// The 8 KB weight table sits in LDS; the undistorted BGR frame is kept for inspection (1 MB).
__global__ __launch_bounds__(256) void k_remap_gray(const uint8_t* __restrict__ src, int rows, int cols, const short2* __restrict__ map1,
                                                     const uint16_t* __restrict__ map2, const short4* __restrict__ wtab,
                                                     uint8_t* __restrict__ undist, uint8_t* __restrict__ gray) {
    __shared__ short4 s_w[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) s_w[i] = wtab[i];
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * cols) return;
    const short2 m = map1[i];
    const short4 w = s_w[map2[i]];
    const int sx = m.x, sy = m.y;
    const bool x0 = (unsigned)sx < (unsigned)cols, x1 = (unsigned)(sx + 1) < (unsigned)cols;
    const bool y0 = (unsigned)sy < (unsigned)rows, y1 = (unsigned)(sy + 1) < (unsigned)rows;
    const uint8_t* p = src + ((ptrdiff_t)sy * cols + sx) * 3;
    int v[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const int a00 = (x0 && y0) ? p[c] : 0, a01 = (x1 && y0) ? p[3 + c] : 0;
        const int a10 = (x0 && y1) ? p[cols * 3 + c] : 0, a11 = (x1 && y1) ? p[cols * 3 + 3 + c] : 0;
        v[c] = min(max((a00 * w.x + a01 * w.y + a10 * w.z + a11 * w.w + (1 << 14)) >> 15, 0), 255);
        undist[(size_t)i * 3 + c] = (uint8_t)v[c];
    }
    gray[i] = (uint8_t)((v[0] * 9798 + v[1] * 19235 + v[2] * 3735 + (1 << 14)) >> 15);
}

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// CLAHE tile histograms (of the reflect-padded frame).  64 tiles would leave three quarters of the CUs idle, so every tile
// is cut into kHistStrips row strips, one block each: LDS histogram, then one global atomic per non-empty bin.
constexpr int kHistStrips = 8;
__global__ __launch_bounds__(256) void k_clahe_hist(const uint8_t* __restrict__ gray, int rows, int cols, int tiles_x, int tw, int th,
                                                     int* __restrict__ hist) {
    __shared__ int s_hist[256];
    const int t = threadIdx.x, tile = blockIdx.x / kHistStrips, strip = blockIdx.x % kHistStrips;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int r0 = th * strip / kHistStrips, r1 = th * (strip + 1) / kHistStrips;
    s_hist[t] = 0;
    __syncthreads();
    for (int i = t; i < tw * (r1 - r0); i += 256) {
        const int y = ty * th + r0 + i / tw, x = tx * tw + i % tw;
        atomicAdd(&s_hist[gray[(size_t)reflect101(y, rows) * cols + reflect101(x, cols)]], 1);
    }
    __syncthreads();
    if (s_hist[t]) atomicAdd(&hist[tile * 256 + t], s_hist[t]);
}

// one block per tile: clip, redistribute, cumulative LUT; leaves the histogram zeroed for the next frame
__global__ __launch_bounds__(256) void k_clahe_lut(int* __restrict__ hist, int clip, float lut_scale, uint8_t* __restrict__ lut) {
    __shared__ int s_scan[256];
    __shared__ int s_clipped;
    const int t = threadIdx.x;
    if (t == 0) s_clipped = 0;
    int h = hist[blockIdx.x * 256 + t];
    hist[blockIdx.x * 256 + t] = 0;
    __syncthreads();
    if (clip > 0) {
        if (h > clip) {
            atomicAdd(&s_clipped, h - clip);
            h = clip;
        }
        __syncthreads();
        const int clipped = s_clipped;
        const int batch = clipped / 256, residual = clipped - batch * 256;
        h += batch;
        if (residual != 0) {
            const int step = max(256 / residual, 1);
            if (t % step == 0 && t / step < residual) h++;
        }
    }
    s_scan[t] = h;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {  // inclusive integer scan
        const int v = t >= d ? s_scan[t - d] : 0;
        __syncthreads();
        s_scan[t] += v;
        __syncthreads();
    }
    lut[(size_t)blockIdx.x * 256 + t] = sat_u8((int)rintf(s_scan[t] * lut_scale));
}

// bilinear blending of the four neighbouring tiles' look-up tables
__global__ __launch_bounds__(256) void k_clahe_apply(const uint8_t* __restrict__ gray, int rows, int cols, int tiles_x, int tiles_y,
                                                      float inv_tw, float inv_th, const uint8_t* __restrict__ lut,
                                                      uint8_t* __restrict__ dst, int dst_step) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= cols) return;
    const float tyf = y * inv_th - 0.5f;
    int ty1 = (int)floorf(tyf), ty2 = ty1 + 1;
    const float ya = tyf - ty1, ya1 = 1.0f - ya;
    ty1 = max(ty1, 0);
    ty2 = min(ty2, tiles_y - 1);
    const float txf = x * inv_tw - 0.5f;
    int tx1 = (int)floorf(txf), tx2 = tx1 + 1;
    const float xa = txf - tx1, xa1 = 1.0f - xa;
    tx1 = max(tx1, 0);
    tx2 = min(tx2, tiles_x - 1);
    const int v = gray[(size_t)y * cols + x];
    const uint8_t* P1 = lut + (size_t)ty1 * tiles_x * 256;
    const uint8_t* P2 = lut + (size_t)ty2 * tiles_x * 256;
    const float res = (P1[tx1 * 256 + v] * xa1 + P1[tx2 * 256 + v] * xa) * ya1 + (P2[tx1 * 256 + v] * xa1 + P2[tx2 * 256 + v] * xa) * ya;
    dst[(size_t)y * dst_step + x] = sat_u8((int)rintf(res));
}

constexpr int kPrepSlots = 2;

}  // namespace

struct vo_prep {
    vo_ctx* ctx = nullptr;
    vo_prep_params p{};
    bool resize = false, fast = false;
    int isx = 1, isy = 1;
    int max_taps_x = 0, max_taps_y = 0;  // most taps of one destination index (generic resize)
    int erows = 0, ecols = 0, tw = 0, th = 0, clip = 0;
    float lut_scale = 0.f;
    // device
    uint8_t* d_src = nullptr;     // [src_rows][src_cols][3]
    uint8_t* d_resized = nullptr; // [dst_rows][dst_cols][3] (aliases d_src when no resize is needed)
    uint8_t* d_undist = nullptr;
    uint8_t* d_gray = nullptr;
    uint8_t* d_out = nullptr;
    uint8_t* d_lut = nullptr;
    int* d_hist = nullptr;        // [tiles][256], zero between frames
    short2* d_map1 = nullptr;
    uint16_t* d_map2 = nullptr;
    short4* d_wtab = nullptr;
    AreaTap *d_xt = nullptr, *d_yt = nullptr;
    int *d_xo = nullptr, *d_yo = nullptr;
    // host
    uint8_t* h_stage[kPrepSlots] = {nullptr, nullptr};  // pinned copies of the incoming frame
    hipEvent_t ev[kPrepSlots] = {nullptr, nullptr};
    bool pending[kPrepSlots] = {false, false};
    int slot = 0;
    uint8_t* h_out = nullptr;  // pinned result of vo_prep_run
};

extern "C" int vo_update_camera_matrix(int src_rows, int src_cols, int dst_rows, int dst_cols, const double K[9], double K_out[9]) {
    if (!K || !K_out || src_rows <= 0 || src_cols <= 0 || dst_rows <= 0 || dst_cols <= 0) return VO_ERR_INVALID_ARG;
    const double RW = ((double)dst_cols) / ((double)src_cols), RH = ((double)dst_rows) / ((double)src_rows);
    // diag(RW, RH, 1) * K through the generic matrix product: every element is a sum of three products, two of them zero
    const double S[9] = {RW, 0, 0, 0, RH, 0, 0, 0, 1};
    double out[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            double s = 0;
            for (int k = 0; k < 3; k++) s += S[i * 3 + k] * K[k * 3 + j];
            out[i * 3 + j] = s;
        }
    std::memcpy(K_out, out, sizeof(out));
    return VO_OK;
}

// initInterTab2D's 15-bit bilinear table.  All products are multiples of 2^-10 and sum to 32768 exactly, except the
// single weight 32768 of cell (0, 0), which a short cannot hold: OpenCV saturates it to 32767 and its sum fix-up, which
// for a 2 x 2 kernel inspects entry [3] of the cell, adds the missing 1 there.
static void build_weight_table(std::vector<int16_t>& tab) {
    tab.assign(1024 * 4, 0);
    for (int c = 0; c < 1024; c++) {
        const float x = (c & 31) * (1.f / 32), y = (c >> 5) * (1.f / 32);
        const float wx[2] = {1.f - x, x}, wy[2] = {1.f - y, y};
        int sum = 0;
        for (int k = 0; k < 4; k++) {
            const int q = (int)lrintf(wy[k >> 1] * wx[k & 1] * 32768.f);
            tab[c * 4 + k] = (int16_t)std::min(q, 32767);
            sum += tab[c * 4 + k];
        }
        if (sum < 32768) tab[c * 4 + 3] = (int16_t)(tab[c * 4 + 3] + (32768 - sum));
    }
}

static int sat_round_i32(double v) {
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return -2147483647 - 1;
    return (int)lrint(v);
}

// cofactor inverse of a 3 x 3 matrix in cv::invert's operation order
static void inverse3(const double a[9], double inv[9]) {
    const double c00 = a[4] * a[8] - a[5] * a[7], c10 = a[3] * a[8] - a[5] * a[6], c20 = a[3] * a[7] - a[4] * a[6];
    double det = a[0] * c00 - a[1] * c10 + a[2] * c20;
    if (det == 0.) {
        std::fill(inv, inv + 9, 0.0);
        return;
    }
    det = 1. / det;
    inv[0] = c00 * det;
    inv[1] = (a[2] * a[7] - a[1] * a[8]) * det;
    inv[2] = (a[1] * a[5] - a[2] * a[4]) * det;
    inv[3] = (a[5] * a[6] - a[3] * a[8]) * det;
    inv[4] = (a[0] * a[8] - a[2] * a[6]) * det;
    inv[5] = (a[2] * a[3] - a[0] * a[5]) * det;
    inv[6] = c20 * det;
    inv[7] = (a[1] * a[6] - a[0] * a[7]) * det;
    inv[8] = (a[0] * a[4] - a[1] * a[3]) * det;
}

// cv::undistort builds its maps in stripes of 4096 / cols rows, each from the inverse of K with the principal point
// moved up by the stripe's first row, and walks along a row by adding the inverse's first column repeatedly.
static void build_undistort_maps(int rows, int cols, const double K[9], const double dist[4], std::vector<int16_t>& map1,
                                 std::vector<uint16_t>& map2) {
    map1.assign((size_t)rows * cols * 2, 0);
    map2.assign((size_t)rows * cols, 0);
    const double fx = K[0], fy = K[4], u0 = K[2], v0 = K[5];
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3];
    const int stripe = std::min(std::max(1, (1 << 12) / std::max(cols, 1)), rows);
    for (int row = 0; row < rows; row++) {
        const int y0 = (row / stripe) * stripe, i = row - y0;
        double Ks[9], ir[9];
        std::copy(K, K + 9, Ks);
        Ks[5] = v0 - y0;
        inverse3(Ks, ir);
        double px = i * ir[1] + ir[2], py = i * ir[4] + ir[5], pw = i * ir[7] + ir[8];
        for (int j = 0; j < cols; j++, px += ir[0], py += ir[3], pw += ir[6]) {
            const double w = 1. / pw, x = px * w, y = py * w;
            const double x2 = x * x, y2 = y * y, r2 = x2 + y2, _2xy = 2 * x * y;
            // k3 .. k6 are zero for the node's four-coefficient model; the polynomial keeps OpenCV's Horner order
            const double kr = (1 + ((0 * r2 + k2) * r2 + k1) * r2) / (1 + ((0 * r2 + 0) * r2 + 0) * r2);
            const double xd = (x * kr + p1 * _2xy + p2 * (r2 + 2 * x2));
            const double yd = (y * kr + p1 * (r2 + 2 * y2) + p2 * _2xy);
            const int iu = sat_round_i32((fx * xd + u0) * 32), iv = sat_round_i32((fy * yd + v0) * 32);
            const size_t o = (size_t)row * cols + j;
            map1[o * 2] = (int16_t)(iu >> 5);
            map1[o * 2 + 1] = (int16_t)(iv >> 5);
            map2[o] = (uint16_t)((iv & 31) * 32 + (iu & 31));
        }
    }
}

template <class T>
static int upload_vec(T** d, const void* h, size_t bytes) {
    VO_HIP(hipMalloc((void**)d, std::max<size_t>(bytes, 16)));
    VO_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
    return VO_OK;
}

extern "C" void vo_prep_destroy(vo_prep* pp) {
    if (!pp) return;
    hipSetDevice(pp->ctx->device);
    hipStreamSynchronize(pp->ctx->stream);
    if (pp->d_resized && pp->d_resized != pp->d_src) hipFree(pp->d_resized);
    void* bufs[] = {pp->d_src, pp->d_undist, pp->d_gray, pp->d_out, pp->d_lut, pp->d_hist, pp->d_map1, pp->d_map2, pp->d_wtab, pp->d_xt, pp->d_yt, pp->d_xo, pp->d_yo};
    for (void* b : bufs)
        if (b) hipFree(b);
    for (int s = 0; s < kPrepSlots; s++) {
        if (pp->h_stage[s]) hipHostFree(pp->h_stage[s]);
        if (pp->ev[s]) hipEventDestroy(pp->ev[s]);
    }
    if (pp->h_out) hipHostFree(pp->h_out);
    delete pp;
}

static int prep_init(vo_prep* pp) {
    const vo_prep_params& p = pp->p;
    const size_t src_bytes = (size_t)p.src_rows * p.src_cols * 3, dst_px = (size_t)p.dst_rows * p.dst_cols;
    VO_HIP(hipMalloc((void**)&pp->d_src, src_bytes));
    if (pp->resize)
        VO_HIP(hipMalloc((void**)&pp->d_resized, dst_px * 3));
    else
        pp->d_resized = pp->d_src;
    VO_HIP(hipMalloc((void**)&pp->d_undist, dst_px * 3));
    VO_HIP(hipMalloc((void**)&pp->d_gray, dst_px));
    VO_HIP(hipMalloc((void**)&pp->d_out, dst_px));
    VO_HIP(hipMalloc((void**)&pp->d_lut, (size_t)p.tiles_x * p.tiles_y * 256));
    VO_HIP(hipMalloc((void**)&pp->d_hist, (size_t)p.tiles_x * p.tiles_y * 256 * sizeof(int)));
    VO_HIP(hipMemset(pp->d_hist, 0, (size_t)p.tiles_x * p.tiles_y * 256 * sizeof(int)));
    for (int s = 0; s < kPrepSlots; s++) {
        VO_HIP(hipHostMalloc((void**)&pp->h_stage[s], src_bytes, hipHostMallocDefault));
        VO_HIP(hipEventCreateWithFlags(&pp->ev[s], hipEventDisableTiming));
    }
    VO_HIP(hipHostMalloc((void**)&pp->h_out, dst_px, hipHostMallocDefault));
    if (pp->resize && !pp->fast) {
        std::vector<AreaTap> xt, yt;
        std::vector<int> xo, yo;
        area_taps(p.src_cols, p.dst_cols, (double)p.src_cols / p.dst_cols, xt, xo);
        area_taps(p.src_rows, p.dst_rows, (double)p.src_rows / p.dst_rows, yt, yo);
        auto most = [](const std::vector<AreaTap>& t, const std::vector<int>& o) {
            int m = 0;
            bool consecutive = true;
            for (size_t d = 0; d + 1 < o.size(); d++) {
                m = std::max(m, o[d + 1] - o[d]);
                for (int k = o[d] + 1; k < o[d + 1]; k++) consecutive &= t[k].si == t[k - 1].si + 1;
            }
            return consecutive ? m : 1 << 20;  // non-consecutive taps (never for a shrink) -> the any-scale kernel
        };
        pp->max_taps_x = most(xt, xo);
        pp->max_taps_y = most(yt, yo);
        VO_TRY(upload_vec(&pp->d_xt, xt.data(), xt.size() * sizeof(AreaTap)));
        VO_TRY(upload_vec(&pp->d_yt, yt.data(), yt.size() * sizeof(AreaTap)));
        VO_TRY(upload_vec(&pp->d_xo, xo.data(), xo.size() * sizeof(int)));
        VO_TRY(upload_vec(&pp->d_yo, yo.data(), yo.size() * sizeof(int)));
    }
    std::vector<int16_t> wtab, map1;
    std::vector<uint16_t> map2;
    build_weight_table(wtab);
    build_undistort_maps(p.dst_rows, p.dst_cols, p.K, p.dist, map1, map2);
    VO_TRY(upload_vec(&pp->d_wtab, wtab.data(), wtab.size() * sizeof(int16_t)));
    VO_TRY(upload_vec(&pp->d_map1, map1.data(), map1.size() * sizeof(int16_t)));
    VO_TRY(upload_vec(&pp->d_map2, map2.data(), map2.size() * sizeof(uint16_t)));
    return VO_OK;
}

extern "C" int vo_prep_create(vo_ctx* ctx, const vo_prep_params* p, vo_prep** out) {
    if (!ctx || !p || !out) return VO_ERR_INVALID_ARG;
    *out = nullptr;
    if (p->src_rows < 16 || p->src_cols < 16 || p->dst_rows < 16 || p->dst_cols < 16 || p->tiles_x < 1 || p->tiles_y < 1 ||
        p->tiles_x > 64 || p->tiles_y > 64 || !(p->clip_limit >= 0.0))
        return VO_ERR_INVALID_ARG;
    if (p->src_rows > 8192 || p->src_cols > 8192) return VO_ERR_UNSUPPORTED;
    // INTER_AREA is the area filter only when shrinking in both directions; OpenCV turns an enlargement into INTER_LINEAR,
    // which the node's cameras never ask for (desired_size 640 x 505 from 2560 x 2056 or less)
    if (p->dst_rows > p->src_rows || p->dst_cols > p->src_cols) return VO_ERR_UNSUPPORTED;
    if (!(p->K[0] != 0.0 && p->K[4] != 0.0)) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(ctx->device));
    vo_prep* pp = new vo_prep();
    pp->ctx = ctx;
    pp->p = *p;
    pp->resize = p->dst_rows != p->src_rows || p->dst_cols != p->src_cols;
    const double scale_x = (double)p->src_cols / p->dst_cols, scale_y = (double)p->src_rows / p->dst_rows;
    pp->isx = (int)lrint(scale_x), pp->isy = (int)lrint(scale_y);
    pp->fast = std::fabs(scale_x - pp->isx) < DBL_EPSILON && std::fabs(scale_y - pp->isy) < DBL_EPSILON;
    pp->erows = p->dst_rows, pp->ecols = p->dst_cols;
    if (!(p->dst_cols % p->tiles_x == 0 && p->dst_rows % p->tiles_y == 0)) {  // CLAHE pads BOTH sides as soon as one needs it
        pp->erows = p->dst_rows + (p->tiles_y - p->dst_rows % p->tiles_y);
        pp->ecols = p->dst_cols + (p->tiles_x - p->dst_cols % p->tiles_x);
    }
    pp->tw = pp->ecols / p->tiles_x, pp->th = pp->erows / p->tiles_y;
    const int tile_total = pp->tw * pp->th;
    pp->lut_scale = (float)(256 - 1) / tile_total;
    pp->clip = 0;
    if (p->clip_limit > 0.0) pp->clip = std::max((int)(p->clip_limit * tile_total / 256), 1);
    const int st = prep_init(pp);
    if (st != VO_OK) {
        vo_prep_destroy(pp);
        return st;
    }
    *out = pp;
    return VO_OK;
}

static int prep_passes(vo_prep* pp, uint8_t* d_dst, int dst_step);

// upload one frame through the pinned staging slots and run the four passes; result to (d_dst, dst_step) on the device
static int prep_enqueue(vo_prep* pp, const uint8_t* bgr, size_t step, uint8_t* d_dst, int dst_step) {
    vo_ctx* ctx = pp->ctx;
    const vo_prep_params& p = pp->p;
    const int s = pp->slot;
    pp->slot = (s + 1) % kPrepSlots;
    if (pp->pending[s]) {
        VO_HIP(hipEventSynchronize(pp->ev[s]));
        pp->pending[s] = false;
    }
    const size_t row_bytes = (size_t)p.src_cols * 3;
    if (step == row_bytes)
        std::memcpy(pp->h_stage[s], bgr, row_bytes * p.src_rows);
    else
        for (int r = 0; r < p.src_rows; r++) std::memcpy(pp->h_stage[s] + r * row_bytes, bgr + (size_t)r * step, row_bytes);
    VO_HIP(hipMemcpyAsync(pp->d_src, pp->h_stage[s], row_bytes * p.src_rows, hipMemcpyHostToDevice, ctx->stream));
    VO_HIP(hipEventRecord(pp->ev[s], ctx->stream));
    pp->pending[s] = true;
    return prep_passes(pp, d_dst, dst_step);
}

// the four passes on the frame resident in d_src
static int prep_passes(vo_prep* pp, uint8_t* d_dst, int dst_step) {
    vo_ctx* ctx = pp->ctx;
    const vo_prep_params& p = pp->p;
    const size_t row_bytes = (size_t)p.src_cols * 3;
    const int R = p.dst_rows, C = p.dst_cols;
    if (pp->resize) {
        StageScope sc(ctx, ST_PREP_RESIZE);
        const dim3 grid((C * 3 + 255) / 256, R);
        const int sb = (int)row_bytes;
        if (pp->fast) {
            auto k = k_resize_area_int<0>;
            if (pp->isx == pp->isy && pp->isx >= 2 && pp->isx <= 4)
                k = pp->isx == 2 ? k_resize_area_int<2> : pp->isx == 3 ? k_resize_area_int<3> : k_resize_area_int<4>;
            k<<<grid, 256, 0, ctx->stream>>>(pp->d_src, sb, pp->isx, pp->isy, pp->d_resized, R, C);
        } else {
            auto k = k_resize_area;
            const int mx = pp->max_taps_x <= 4 ? 4 : pp->max_taps_x <= 6 ? 6 : 8, my = pp->max_taps_y <= 4 ? 4 : pp->max_taps_y <= 6 ? 6 : 8;
            // staged form: 86 destination pixels of at most 7 source columns each fit one LDS row
            const bool lds = sb % 16 == 0 && pp->max_taps_y <= 8 && pp->max_taps_x <= 8 && (double)p.src_cols / p.dst_cols <= 7.0;
            if (lds) {
                switch (mx * 10 + my) {
                    case 44: k = k_resize_area_lds<4, 4>; break;
                    case 46: k = k_resize_area_lds<4, 6>; break;
                    case 48: k = k_resize_area_lds<4, 8>; break;
                    case 64: k = k_resize_area_lds<6, 4>; break;
                    case 66: k = k_resize_area_lds<6, 6>; break;
                    case 68: k = k_resize_area_lds<6, 8>; break;
                    case 84: k = k_resize_area_lds<8, 4>; break;
                    case 86: k = k_resize_area_lds<8, 6>; break;
                    default: k = k_resize_area_lds<8, 8>; break;
                }
            } else if (pp->max_taps_x <= 8 && pp->max_taps_y <= 8) {
                switch (mx * 10 + my) {
                    case 44: k = k_resize_area_u<4, 4>; break;
                    case 46: k = k_resize_area_u<4, 6>; break;
                    case 48: k = k_resize_area_u<4, 8>; break;
                    case 64: k = k_resize_area_u<6, 4>; break;
                    case 66: k = k_resize_area_u<6, 6>; break;
                    case 68: k = k_resize_area_u<6, 8>; break;
                    case 84: k = k_resize_area_u<8, 4>; break;
                    case 86: k = k_resize_area_u<8, 6>; break;
                    default: k = k_resize_area_u<8, 8>; break;
                }
            }
            k<<<grid, 256, 0, ctx->stream>>>(pp->d_src, sb, pp->d_xt, pp->d_xo, pp->d_yt, pp->d_yo, pp->d_resized, R, C);
        }
    }
    {
        StageScope sc(ctx, ST_PREP_REMAP);
        k_remap_gray<<<(R * C + 255) / 256, 256, 0, ctx->stream>>>(pp->d_resized, R, C, pp->d_map1, pp->d_map2, pp->d_wtab, pp->d_undist, pp->d_gray);
    }
    {
        StageScope sc(ctx, ST_PREP_CLAHE);
        k_clahe_hist<<<p.tiles_x * p.tiles_y * kHistStrips, 256, 0, ctx->stream>>>(pp->d_gray, R, C, p.tiles_x, pp->tw, pp->th, pp->d_hist);
        k_clahe_lut<<<p.tiles_x * p.tiles_y, 256, 0, ctx->stream>>>(pp->d_hist, pp->clip, pp->lut_scale, pp->d_lut);
        k_clahe_apply<<<dim3((C + 255) / 256, R), 256, 0, ctx->stream>>>(pp->d_gray, R, C, p.tiles_x, p.tiles_y, 1.0f / pp->tw, 1.0f / pp->th,
                                                                          pp->d_lut, d_dst, dst_step);
    }
    VO_HIP(hipGetLastError());
    return VO_OK;
}

extern "C" int vo_prep_run(vo_prep* pp, const uint8_t* bgr, size_t step, uint8_t* gray, size_t gray_step) {
    if (!pp || !bgr || !gray || step < (size_t)pp->p.src_cols * 3 || gray_step < (size_t)pp->p.dst_cols) return VO_ERR_INVALID_ARG;
    vo_ctx* ctx = pp->ctx;
    VO_HIP(hipSetDevice(ctx->device));
    VO_TRY(prep_enqueue(pp, bgr, step, pp->d_out, pp->p.dst_cols));
    const size_t px = (size_t)pp->p.dst_rows * pp->p.dst_cols;
    VO_HIP(hipMemcpyAsync(pp->h_out, pp->d_out, px, hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    for (int r = 0; r < pp->p.dst_rows; r++) std::memcpy(gray + (size_t)r * gray_step, pp->h_out + (size_t)r * pp->p.dst_cols, pp->p.dst_cols);
    return VO_OK;
}

extern "C" int vo_prep_to_batch(vo_prep* pp, const uint8_t* bgr, size_t step, vo_batch* b, int frame) {
    if (!pp || !bgr || !b || step < (size_t)pp->p.src_cols * 3) return VO_ERR_INVALID_ARG;
    if (b->ctx != pp->ctx || frame < 0 || frame >= b->n_frames) return VO_ERR_INVALID_ARG;
    if (b->rows != pp->p.dst_rows || b->cols != pp->p.dst_cols) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(pp->ctx->device));
    // same stream as the batch's pipeline: the frame is in place before the kernels of the next vo_batch_run start
    uint8_t* dst = nullptr;
    VO_TRY(vo_batch_frame_slot(b, frame, &dst));
    return prep_enqueue(pp, bgr, step, dst, b->cols);
}

extern "C" int vo_prep_rerun(vo_prep* pp, int reps) {
    if (!pp || reps < 0) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(pp->ctx->device));
    for (int r = 0; r < reps; r++) VO_TRY(prep_passes(pp, pp->d_out, pp->p.dst_cols));
    return VO_OK;
}

extern "C" int vo_prep_fetch(vo_prep* pp, int what, void* dst, size_t bytes) {
    if (!pp || !dst) return VO_ERR_INVALID_ARG;
    const size_t px = (size_t)pp->p.dst_rows * pp->p.dst_cols;
    const void* src = nullptr;
    size_t n = 0;
    switch (what) {
        case 0: src = pp->d_resized, n = px * 3; break;
        case 1: src = pp->d_undist, n = px * 3; break;
        case 2: src = pp->d_gray, n = px; break;
        case 3: src = pp->d_lut, n = (size_t)pp->p.tiles_x * pp->p.tiles_y * 256; break;
        case 4: src = pp->d_map1, n = px * 4; break;
        case 5: src = pp->d_map2, n = px * 2; break;
        default: return VO_ERR_INVALID_ARG;
    }
    if (bytes != n) return VO_ERR_INVALID_ARG;
    VO_HIP(hipSetDevice(pp->ctx->device));
    VO_HIP(hipStreamSynchronize(pp->ctx->stream));
    VO_HIP(hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    return VO_OK;
}
