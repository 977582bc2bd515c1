// SURF detect + describe for gfx950.
//
// Replaces SURF::create(50,4,3,false,false)->detectAndCompute(img, noArray(), kps, desc)
// (reference: include/visual_odometry/camera_utility.hpp:746-750).
//
// Kernels (all batched over frames):
//   k_colsum, k_colsum_scan, k_integral : u8 image -> int32 integral image, banded two-level scan
//   k_hessian_layer                    : det/trace of one layer (inspection entry point only)
//   k_hessian_nms_fused / _lds         : box-filter Hessian of the 5 layers of an octave for one sample tile, integral
//                                        patch staged in LDS (octaves 0 and 1; taps from L1/L2 for octaves 2-3), two-pass
//                                        3x3x3 non-max suppression with an LDS candidate queue, sub-sample interpolation,
//                                        wavefront-ballot compaction; only the keypoint list is written to HBM
//   k_sort_bitonic (k_rank + k_scatter) : KeypointGreater order, bitonic in LDS (rank sort above 16384 keypoints)
//   k_orientation                      : dominant orientation, two keypoints per wavefront
//   k_quads, k_classify_*              : 2x2 footprint image; window-class work lists sorted by (frame, tile)
//   k_describe_tiled<>                 : 64-float descriptor; rotated window sampled in compact 2-D gather blocks from
//                                        exact closed-form coordinates, INTER_AREA decimation streamed through LDS
//   k_describe<>                       : the reference's sequential walk, fallback for the rare inexact windows
// Arithmetic follows the order of operations of SURVEY.md Appendix A exactly; the file is
// compiled with -ffp-contract=off so that every float/double operation rounds once.
#include "vo_common.hpp"
#include <type_traits>

#include <cfloat>
#include <cmath>

__constant__ DetectTables c_tab;

// ------------------------------------------------------------------------------------------------
// host: tables
// ------------------------------------------------------------------------------------------------
static inline int h_cvRoundF(float v) { return (int)lrintf(v); }

static void gaussian_kernel_f32(int n, double sigma, float* cf) {
    // cv::getGaussianKernel(n, sigma, CV_32F), non-tabulated branch
    double scale2X = -0.5 / (sigma * sigma), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        cf[i] = (float)std::exp(scale2X * x * x);
        sum += cf[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) cf[i] = (float)(cf[i] * sum);
}

// Box corners of one layer: cvRound(size/9 * v).  size*v/9 is never within 1/18 of a tie, so the
// integer form below equals the float rounding of the reference; vo_build_detect_tables checks that.
struct BoxGeom {
    int ex[4];   // R(0), R(3), R(6), R(9): x edges of Dxx boxes = y edges of Dyy boxes
    int e27[2];  // R(2), R(7)
    int c[4];    // R(1), R(4), R(5), R(8): Dxy corners
};
constexpr int box_round(int size, int v) { return (2 * size * v + 9) / 18; }
constexpr BoxGeom box_geom(int size) {
    return BoxGeom{{box_round(size, 0), box_round(size, 3), box_round(size, 6), box_round(size, 9)},
                   {box_round(size, 2), box_round(size, 7)},
                   {box_round(size, 1), box_round(size, 4), box_round(size, 5), box_round(size, 8)}};
}

// The ten box weights of a layer, w = sign / ((float)(dx2 - dx1) * (dy2 - dy1)) as vo_build_detect_tables computes them (which
// checks that the two agree).  Compile-time for the LDS-tile kernels: a weight is then a literal operand of its multiply - at the
// FP32 rate, where an SGPR operand halves it - and, what matters more, no scalar load of the constant table sits in the det
// loop: scalar loads and LDS reads share one wait counter, and every s_waitcnt for a weight drained the taps in flight.
struct BoxWeights {
    float w[10];
};
constexpr BoxWeights box_weights(int size) {
    const BoxGeom G = box_geom(size);
    BoxWeights r{};
    const float wsgn[3] = {1, -2, 1};
    for (int k = 0; k < 3; k++) {
        r.w[k] = wsgn[k] / ((float)(G.ex[k + 1] - G.ex[k]) * (float)(G.e27[1] - G.e27[0]));
        r.w[3 + k] = wsgn[k] / ((float)(G.e27[1] - G.e27[0]) * (float)(G.ex[k + 1] - G.ex[k]));
    }
    const int bx[4] = {0, 2, 0, 2}, by[4] = {0, 0, 2, 2};
    const float ws[4] = {1, -1, -1, 1};
    for (int k = 0; k < 4; k++) r.w[6 + k] = ws[k] / ((float)(G.c[bx[k] + 1] - G.c[bx[k]]) * (float)(G.c[by[k] + 1] - G.c[by[k]]));
    return r;
}

// Box pattern of one layer of the given size (resizeHaarPattern): corners cvRound(size/9 * v) in float, weights
// sign / ((float)(dx2 - dx1) * (dy2 - dy1)).  Any size: the n_octave_layers != 3 path builds its layers with it too.
static LayerTab make_layer_tab(int size, int step) {
    LayerTab L;
    std::memset(&L, 0, sizeof(L));
    L.size = size;
    L.step = step;
    L.margin = (size / 2) / step;
    const float ratio = (float)size / 9;
    auto R = [&](int v) { return h_cvRoundF(ratio * v); };
    const int ex[4] = {R(0), R(3), R(6), R(9)};
    const int e27[2] = {R(2), R(7)};
    const int c[4] = {R(1), R(4), R(5), R(8)};
    for (int k = 0; k < 4; k++) {
        L.xdx[k] = ex[k];
        L.ydy[k] = ex[k];
        L.cxy[k] = c[k];
    }
    for (int k = 0; k < 2; k++) {
        L.ydx[k] = e27[k];
        L.xdy[k] = e27[k];
    }
    const float wsgn[3] = {1, -2, 1};
    for (int k = 0; k < 3; k++) {
        // w = src[4] / ((float)(dx2-dx1) * (dy2-dy1))
        L.w[k] = wsgn[k] / ((float)(ex[k + 1] - ex[k]) * (e27[1] - e27[0]));
        L.w[3 + k] = wsgn[k] / ((float)(e27[1] - e27[0]) * (ex[k + 1] - ex[k]));
    }
    const int bx[4] = {0, 2, 0, 2}, by[4] = {0, 0, 2, 2};
    const float ws[4] = {1, -1, -1, 1};
    for (int k = 0; k < 4; k++) L.w[6 + k] = ws[k] / ((float)(c[bx[k] + 1] - c[bx[k]]) * (c[by[k] + 1] - c[by[k]]));
    return L;
}

void vo_build_detect_tables(DetectTables& t, int) {
    std::memset(&t, 0, sizeof(t));
    for (int o = 0; o < kNumOctaves; o++)
        for (int l = 0; l < kLayersPerOctave; l++) {
            LayerTab& L = t.layers[o * kLayersPerOctave + l];
            L = make_layer_tab((9 + 6 * l) << o, 1 << o);
            {   // the integer form used for compile-time box corners (box_geom) must agree with the float rounding
                const float ratio = (float)L.size / 9;
                for (int k = 0; k < 10; k++)
                    if (h_cvRoundF(ratio * k) != (2 * L.size * k + 9) / 18) {
                        std::fprintf(stderr, "vo_mi355x: box corner table mismatch (size %d, v %d)\n", L.size, k);
                        std::abort();
                    }
            }
            const BoxWeights cw = box_weights(L.size);  // what the LDS-tile kernels use as literals
            if (std::memcmp(cw.w, L.w, sizeof(cw.w)) != 0) {
                std::fprintf(stderr, "vo_mi355x: box weight table mismatch (size %d)\n", L.size);
                std::abort();
            }
        }
    float G_ori[13];
    gaussian_kernel_f32(13, 2.5, G_ori);
    int n = 0;
    for (int i = -6; i <= 6; i++)
        for (int j = -6; j <= 6; j++)
            if (i * i + j * j <= 36) {
                t.aptx[n] = i;
                t.apty[n] = j;
                t.aptw[n++] = G_ori[i + 6] * G_ori[j + 6];
            }
    float G_desc[kPatch];
    gaussian_kernel_f32(kPatch, (double)3.3f, G_desc);  // SURF_DESC_SIGMA is a float constant
    for (int i = 0; i < kPatch; i++)
        for (int j = 0; j < kPatch; j++) t.DW[i * kPatch + j] = G_desc[i] * G_desc[j];
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int d_cvRoundF(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int d_cvRoundD(double v) { return __double2int_rn(v); }
__device__ __forceinline__ int d_cvFloorD(double v) {
    int i = (int)v;
    return i - (i > v);
}
__device__ __forceinline__ int d_cvCeilD(double v) {
    int i = (int)v;
    return i + (i < v);
}

// cv::fastAtan2 (degrees)
__device__ __forceinline__ float d_fastAtan2(float y, float x) {
    const float scale = (float)(180 / 3.1415926535897932384626433832795);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    // the reference's two branches (ax >= ay: ay / (ax + eps); else ax / (ay + eps), a = 90 - ...) with ONE division:
    // the operands are selected, the arithmetic is the same (a divergent if/else runs the division sequence twice)
    const bool steep = !(ax >= ay);
    const float num = steep ? ax : ay, den = (steep ? ay : ax) + (float)DBL_EPSILON;
    const float c = num / den;
    const float c2 = c * c;
    float a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    if (steep) a = 90.f - a;
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Dx, Dy, Dxy of one layer at one sample.  tap(y, x) returns the integral image at offset (y, x) from the
// sample's box origin.  Integer box sums are exact in any association (mod 2^32); the float multiply and the
// double accumulation follow calcHaarPattern.
template <class Tap>
__device__ __forceinline__ void hessian_boxes_t(const BoxGeom& G, const float* __restrict__ w, const Tap& tap, float& dx,
                                                float& dy, float& dxy) {
    {
        const int y0 = G.e27[0], y1 = G.e27[1];
        uint32_t v0 = tap(y1, G.ex[0]) - tap(y0, G.ex[0]);
        uint32_t v1 = tap(y1, G.ex[1]) - tap(y0, G.ex[1]);
        uint32_t v2 = tap(y1, G.ex[2]) - tap(y0, G.ex[2]);
        uint32_t v3 = tap(y1, G.ex[3]) - tap(y0, G.ex[3]);
        double d = (double)((float)(int32_t)(v1 - v0) * w[0]);
        d += (double)((float)(int32_t)(v2 - v1) * w[1]);
        d += (double)((float)(int32_t)(v3 - v2) * w[2]);
        dx = (float)d;
    }
    {
        const int x0 = G.e27[0], x1 = G.e27[1];
        uint32_t h0 = tap(G.ex[0], x1) - tap(G.ex[0], x0);
        uint32_t h1 = tap(G.ex[1], x1) - tap(G.ex[1], x0);
        uint32_t h2 = tap(G.ex[2], x1) - tap(G.ex[2], x0);
        uint32_t h3 = tap(G.ex[3], x1) - tap(G.ex[3], x0);
        double d = (double)((float)(int32_t)(h1 - h0) * w[3]);
        d += (double)((float)(int32_t)(h2 - h1) * w[4]);
        d += (double)((float)(int32_t)(h3 - h2) * w[5]);
        dy = (float)d;
    }
    {
        const int c0 = G.c[0], c1 = G.c[1], c2 = G.c[2], c3 = G.c[3];
        // horizontal differences over [c0,c1] and [c2,c3] on the four rows
        uint32_t a0 = tap(c0, c1) - tap(c0, c0), b0 = tap(c0, c3) - tap(c0, c2);
        uint32_t a1 = tap(c1, c1) - tap(c1, c0), b1 = tap(c1, c3) - tap(c1, c2);
        uint32_t a2 = tap(c2, c1) - tap(c2, c0), b2 = tap(c2, c3) - tap(c2, c2);
        uint32_t a3 = tap(c3, c1) - tap(c3, c0), b3 = tap(c3, c3) - tap(c3, c2);
        double d = (double)((float)(int32_t)(a1 - a0) * w[6]);  // x[c0,c1] y[c0,c1]
        d += (double)((float)(int32_t)(b1 - b0) * w[7]);        // x[c2,c3] y[c0,c1]
        d += (double)((float)(int32_t)(a3 - a2) * w[8]);        // x[c0,c1] y[c2,c3]
        d += (double)((float)(int32_t)(b3 - b2) * w[9]);        // x[c2,c3] y[c2,c3]
        dxy = (float)d;
    }
}

struct GlobalTap {  // integral image in HBM / L2, row pitch sw
    const int32_t* __restrict__ org;
    int sw;
    __device__ __forceinline__ uint32_t operator()(int y, int x) const { return (uint32_t)org[y * sw + x]; }
};

__device__ __forceinline__ void hessian_boxes(const LayerTab& L, const int32_t* __restrict__ org, int sw, float& dx,
                                              float& dy, float& dxy) {
    const BoxGeom G{{L.xdx[0], L.xdx[1], L.xdx[2], L.xdx[3]}, {L.ydx[0], L.ydx[1]}, {L.cxy[0], L.cxy[1], L.cxy[2], L.cxy[3]}};
    hessian_boxes_t(G, L.w, GlobalTap{org, sw}, dx, dy, dxy);
}

// Column-phase planes of the integral image (k_sum_planes): sum[y][x] at plane x & 3, row y, word x >> 2.  Octave o >= 2 samples
// every 4th / 8th pixel: for one box corner the lanes of a tile row read words that are 1 / 2 apart in ONE plane - two or four
// cache lines per wavefront load where the row-major image gives eight or sixteen (a quarter / an eighth of each used), which
// is what bounded those tiles (L2 -> L1 line fills).  `lane` points at plane 0, row = the sample's box origin row, word = origin
// column / 4 (a multiple of 4); the rest of a tap's address is the same in every lane.
struct PlanarTap {
    const int32_t* __restrict__ plane0;
    uint32_t lane;  // BYTE offset of this lane's box origin (a frame's planes are < 4 GB): scalar base + 32-bit vector offset
    int pw;         // words per plane row
    size_t ps;      // words per plane
    __device__ __forceinline__ uint32_t operator()(int y, int x) const {
        const int32_t* __restrict__ u = plane0 + ((size_t)(x & 3) * ps + (size_t)y * pw + (x >> 2));
        return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(u) + lane);
    }
};

__device__ __forceinline__ void hessian_boxes_planar(const LayerTab& L, const PlanarTap& tap, float& dx, float& dy, float& dxy) {
    const BoxGeom G{{L.xdx[0], L.xdx[1], L.xdx[2], L.xdx[3]}, {L.ydx[0], L.ydx[1]}, {L.cxy[0], L.cxy[1], L.cxy[2], L.cxy[3]}};
    hessian_boxes_t(G, L.w, tap, dx, dy, dxy);
}

__device__ __forceinline__ float hessian_det(float dx, float dy, float dxy) {
    float t1 = dx * dy;
    float t2 = 0.81f * dxy;
    float t3 = t2 * dxy;
    return t1 - t3;
}

// ------------------------------------------------------------------------------------------------
// a1: integral image.  Bands of 16 rows; k_colsum gives each band's column sums, k_integral
// carries them in and scans 16 rows x W with one workgroup per band.
// ------------------------------------------------------------------------------------------------
constexpr int kBand = 16;

__global__ __launch_bounds__(256) void k_colsum(const uint8_t* __restrict__ img, size_t img_stride, int H, int W,
                                                int nbands, int32_t* __restrict__ colsum) {
    const int frame = blockIdx.z, band = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const uint8_t* p = img + (size_t)frame * img_stride + (size_t)band * kBand * W + x;
    const int rows = min(kBand, H - band * kBand);
    int s = 0;
    for (int r = 0; r < rows; r++) s += p[(size_t)r * W];
    colsum[((size_t)frame * nbands + band) * W + x] = s;
}

// four columns per lane from aligned words (W and the frame pitch multiples of four); the grid is flat over (band, column
// quad) so that no lane idles at a row's end, and a full band's 16 loads go out back to back
__global__ __launch_bounds__(256) void k_colsum4(const uint8_t* __restrict__ img, size_t img_stride, int H, int W,
                                                 int nbands, int32_t* __restrict__ colsum) {
    const int frame = blockIdx.z, wq = W >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= nbands * wq) return;
    const int band = idx / wq, x = (idx - band * wq) * 4;
    const uint8_t* p = img + (size_t)frame * img_stride + (size_t)band * kBand * W + x;
    const int rows = min(kBand, H - band * kBand);
    // two columns per accumulator: 16 rows x 255 fit 16 bits
    uint32_t even = 0, odd = 0;
    if (rows == kBand) {
        uint32_t v[kBand];
#pragma unroll
        for (int r = 0; r < kBand; r++) v[r] = *reinterpret_cast<const uint32_t*>(p + (size_t)r * W);
#pragma unroll
        for (int r = 0; r < kBand; r++) {
            even += v[r] & 0x00ff00ffu;
            odd += (v[r] >> 8) & 0x00ff00ffu;
        }
    } else {
        for (int r = 0; r < rows; r++) {
            const uint32_t v = *reinterpret_cast<const uint32_t*>(p + (size_t)r * W);
            even += v & 0x00ff00ffu;
            odd += (v >> 8) & 0x00ff00ffu;
        }
    }
    const int4 o = make_int4((int)(even & 0xffffu), (int)(odd & 0xffffu), (int)(even >> 16), (int)(odd >> 16));
    *reinterpret_cast<int4*>(colsum + ((size_t)frame * nbands + band) * W + x) = o;
}

// exclusive prefix of the band sums down every column (in place)
__global__ __launch_bounds__(256) void k_colsum_scan(int W, int nbands, int32_t* __restrict__ colsum) {
    const int frame = blockIdx.y;
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    int32_t* p = colsum + (size_t)frame * nbands * W + x;
    int32_t run = 0;
    int b = 0;
    for (; b + 8 <= nbands; b += 8) {  // eight loads in flight
        int32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = p[(size_t)(b + u) * W];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            p[(size_t)(b + u) * W] = run;
            run += v[u];
        }
    }
    for (; b < nbands; b++) {
        const int32_t v = p[(size_t)b * W];
        p[(size_t)b * W] = run;
        run += v;
    }
}

template <int CPT>
__global__ __launch_bounds__(1024) void k_integral(const uint8_t* __restrict__ img, size_t img_stride, int H, int W,
                                                   int nbands, const int32_t* __restrict__ colsum,
                                                   int32_t* __restrict__ sum, size_t sum_stride) {
    __shared__ int32_t s_wtot[kBand][16];
    const int frame = blockIdx.y, band = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sw = W + 1;
    const uint8_t* im = img + (size_t)frame * img_stride;
    int32_t* out = sum + (size_t)frame * sum_stride;
    const int32_t* cs = colsum + (size_t)frame * nbands * W;
    const int x0 = tid * CPT;
    // column sums of all rows above this band (k_colsum_scan made them exclusive prefixes over the bands)
    int32_t carry[CPT];
#pragma unroll
    for (int c = 0; c < CPT; c++) carry[c] = (x0 + c < W) ? cs[(size_t)band * W + x0 + c] : 0;
    if (band == 0)
        for (int x = tid; x < sw; x += 1024) out[x] = 0;
    int32_t pre[kBand][CPT];  // inclusive prefix over the thread's columns, per row
    int32_t excl[kBand];      // exclusive prefix of thread totals within the wave
#pragma unroll
    for (int r = 0; r < kBand; r++) {
        const int y = band * kBand + r;
        int32_t run = 0;
#pragma unroll
        for (int c = 0; c < CPT; c++) {
            if (y < H && x0 + c < W) carry[c] += im[(size_t)y * W + x0 + c];
            run += (x0 + c < W) ? carry[c] : 0;
            pre[r][c] = run;
        }
        int32_t inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int32_t v = __shfl_up(inc, d, 64);
            if (lane >= d) inc += v;
        }
        excl[r] = inc - run;
        if (lane == 63) s_wtot[r][wave] = inc;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kBand; r++) {
        const int y = band * kBand + r;
        if (y >= H) break;
        int32_t off = excl[r];
        for (int w = 0; w < wave; w++) off += s_wtot[r][w];
        int32_t* row = out + (size_t)(y + 1) * sw;
        if (tid == 0) row[0] = 0;
#pragma unroll
        for (int c = 0; c < CPT; c++)
            if (x0 + c < W) row[x0 + c + 1] = pre[r][c] + off;
    }
}

// Inclusive prefix sum over the 64 lanes of a wavefront with data-parallel-primitive adds (no LDS traffic, unlike the
// ds_bpermute behind __shfl_up): four shifts inside the 16-lane rows, then lane 15 of every row into the next row (rows 1
// and 3), then lane 31 into rows 2 and 3.  Lanes that a step does not write add the `old` operand, 0.
__device__ __forceinline__ int wave_incl_scan(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, true);  // row_bcast:15, rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, true);  // row_bcast:31, rows 2 and 3
    return v;
}

struct __attribute__((packed, aligned(4))) Int4U {  // four consecutive ints at a 4-byte aligned address (sum rows have an odd pitch)
    int v[4];
};

// Frames whose rows are 4-byte aligned (W and the frame pitch multiples of four): four columns per lane from ONE aligned word
// per row (the byte loads of k_integral<CPT> run at the texture addresser's 4 lanes per clock: 2.3 TB/s), the row scan on DPP
// adds, one 16-byte store per lane and row.  A workgroup takes one 16-row band; frames wider than NT * 4 columns go through
// it in column chunks, the row sums carried from chunk to chunk.
template <int NT, bool MULTI>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_integral4(const uint8_t* __restrict__ img, size_t img_stride, int H, int W, int nbands,
                                                  const int32_t* __restrict__ colsum, int32_t* __restrict__ sum, size_t sum_stride,
                                                  int32_t* __restrict__ sum4, int pw, size_t ps) {
    constexpr int NWV = NT / 64;
    __shared__ int32_t s_wtot[kBand][NWV + 1];  // per row: the wavefronts' totals, then (after the second-level scan) their
                                                // exclusive prefixes and, in the last slot, the row's chunk total
    static_assert(kBand * NWV <= NT, "second-level scan: one lane per (row, wavefront)");
    const int frame = blockIdx.y, band = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sw = W + 1;
    const uint8_t* im = img + (size_t)frame * img_stride;
    int32_t* out = sum + (size_t)frame * sum_stride;
    // the same values a second time, by column phase (PlanarTap): the lane's four columns x + 1 .. x + 4 go to planes 1, 2, 3, 0
    int32_t* out4 = sum4 + (size_t)frame * 4 * ps;
    const int32_t* cs = colsum + ((size_t)frame * nbands + band) * W;
    if (band == 0) {
        for (int x = tid; x < sw; x += NT) out[x] = 0;
        for (int x = tid; x < 4 * pw; x += NT) out4[(size_t)(x / pw) * ps + (x % pw)] = 0;
    }
    int32_t chunk_carry[MULTI ? kBand : 1];  // sum of the row's pixels left of the current chunk (frames wider than NT * 4)
#pragma unroll
    for (int r = 0; r < (MULTI ? kBand : 1); r++) chunk_carry[r] = 0;
    for (int xc = 0; xc < (MULTI ? W : 1); xc += NT * 4) {
        const int x = xc + tid * 4;
        const bool live = x < W;
        // column sums of all rows above this band (k_colsum_scan made them exclusive prefixes over the bands)
        int4 carry = make_int4(0, 0, 0, 0);
        if (live) carry = *reinterpret_cast<const int4*>(cs + x);
        uint32_t px[kBand];
#pragma unroll
        for (int r = 0; r < kBand; r++) {
            const int y = band * kBand + r;
            px[r] = (live && y < H) ? *reinterpret_cast<const uint32_t*>(im + (size_t)y * W + x) : 0u;
        }
        // pass 1: the lane's row totals and their scan.  Only the packed pixels and the scan results stay in registers (the
        // kernel has to fit beside the geometry kernels of the previous run, which fill most of the register files); pass 2
        // re-runs the cheap column adds from the saved carry.
        const int4 carry0 = carry;
        int32_t excl[kBand];  // exclusive prefix of the lanes' totals within the wavefront
#pragma unroll
        for (int r = 0; r < kBand; r++) {
            carry.x += (int)(px[r] & 255u);
            carry.y += (int)((px[r] >> 8) & 255u);
            carry.z += (int)((px[r] >> 16) & 255u);
            carry.w += (int)(px[r] >> 24);
            const int tot = live ? (carry.x + carry.y) + (carry.z + carry.w) : 0;
            const int inc = wave_incl_scan(tot);
            excl[r] = inc - tot;
            if (lane == 63) s_wtot[r][wave] = inc;
        }
        __syncthreads();
        if (tid < kBand) {  // second-level scan: one lane per row walks the row's NWV totals
            int32_t run = 0;
#pragma unroll
            for (int w = 0; w < NWV; w++) {
                const int32_t t = s_wtot[tid][w];
                s_wtot[tid][w] = run;
                run += t;
            }
            s_wtot[tid][NWV] = run;
        }
        __syncthreads();
        carry = carry0;
#pragma unroll
        for (int r = 0; r < kBand; r++) {
            const int y = band * kBand + r;
            carry.x += (int)(px[r] & 255u);
            carry.y += (int)((px[r] >> 8) & 255u);
            carry.z += (int)((px[r] >> 16) & 255u);
            carry.w += (int)(px[r] >> 24);
            int32_t off = excl[r] + s_wtot[r][wave];
            if (MULTI) {
                off += chunk_carry[r];
                chunk_carry[r] += s_wtot[r][NWV];
            }
            if (y < H) {
                int32_t* row = out + (size_t)(y + 1) * sw;
                int32_t* row4 = out4 + (size_t)(y + 1) * pw + (x >> 2);
                if (xc == 0 && tid == 0) row[0] = 0, row4[0] = 0;
                if (live) {
                    Int4U o;
                    o.v[0] = carry.x + off;
                    o.v[1] = o.v[0] + carry.y;
                    o.v[2] = o.v[1] + carry.z;
                    o.v[3] = o.v[2] + carry.w;
                    *reinterpret_cast<Int4U*>(row + x + 1) = o;
                    row4[ps] = o.v[0];
                    row4[2 * ps] = o.v[1];
                    row4[3 * ps] = o.v[2];
                    row4[1] = o.v[3];
                }
            }
        }
        __syncthreads();  // s_wtot is rewritten by the next chunk
    }
}

// Column-phase copy of the integral image for the coarse octaves (PlanarTap): plane p of a frame holds columns p, p + 4, ...
__global__ __launch_bounds__(256) void k_sum_planes(const int32_t* __restrict__ sum_base, size_t sum_stride, int H, int W,
                                                    int32_t* __restrict__ out_base, int pw, size_t ps) {
    const int y = blockIdx.y, f = blockIdx.z, jj = blockIdx.x * 256 + threadIdx.x;
    const int sw = W + 1;
    if (4 * jj > W) return;
    const int32_t* __restrict__ row = sum_base + (size_t)f * sum_stride + (size_t)y * sw;
    int32_t v[4];
    if (4 * jj + 3 <= W) {
        const Int4U q = *reinterpret_cast<const Int4U*>(row + 4 * jj);
        v[0] = q.v[0], v[1] = q.v[1], v[2] = q.v[2], v[3] = q.v[3];
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = 4 * jj + k <= W ? row[4 * jj + k] : 0;
    }
    int32_t* __restrict__ out = out_base + (size_t)f * 4 * ps + (size_t)y * pw + jj;
#pragma unroll
    for (int k = 0; k < 4; k++) out[(size_t)k * ps] = v[k];
}

static int enqueue_integral_rows(vo_ctx* ctx, FrameSet& fs, hipStream_t stream);

int vo_enqueue_integral(vo_ctx* ctx, FrameSet& fs, hipStream_t stream) {
    VO_TRY(enqueue_integral_rows(ctx, fs, stream));
    if (fs.W % 4 == 0 && fs.img_stride() % 4 == 0) return VO_OK;  // k_integral4 wrote the planes itself
    StageScope sc(ctx, ST_INTEGRAL, stream);
    const int pw = fs.sum4_pitch();
    hipLaunchKernelGGL(k_sum_planes, dim3(((fs.W + 4) / 4 + 255) / 256, fs.H + 1, fs.F), dim3(256), 0, stream, fs.sum, fs.sum_stride(), fs.H,
                       fs.W, fs.sum4, pw, fs.sum4_plane());
    VO_HIP(hipGetLastError());
    return VO_OK;
}

static int enqueue_integral_rows(vo_ctx* ctx, FrameSet& fs, hipStream_t stream) {
    StageScope sc(ctx, ST_INTEGRAL, stream);
    const int nbands = (fs.H + kBand - 1) / kBand;
    dim3 g1((fs.W + 255) / 256, nbands, fs.F);
    const bool aligned4 = fs.W % 4 == 0 && fs.img_stride() % 4 == 0;
    if (aligned4)
        hipLaunchKernelGGL(k_colsum4, dim3((nbands * (fs.W / 4) + 255) / 256, 1, fs.F), dim3(256), 0, stream, fs.img, fs.img_stride(),
                           fs.H, fs.W, nbands, fs.colsum);
    else
        hipLaunchKernelGGL(k_colsum, g1, dim3(256), 0, stream, fs.img, fs.img_stride(), fs.H, fs.W, nbands, fs.colsum);
    hipLaunchKernelGGL(k_colsum_scan, dim3((fs.W + 255) / 256, fs.F), dim3(256), 0, stream, fs.W, nbands, fs.colsum);
    dim3 g2(nbands, fs.F);
    if (aligned4) {
        if (fs.W <= 1024)
            hipLaunchKernelGGL((k_integral4<256, false>), g2, dim3(256), 0, stream, fs.img, fs.img_stride(), fs.H, fs.W, nbands, fs.colsum,
                               fs.sum, fs.sum_stride(), fs.sum4, fs.sum4_pitch(), fs.sum4_plane());
        else if (fs.W <= 2048)
            hipLaunchKernelGGL((k_integral4<512, false>), g2, dim3(512), 0, stream, fs.img, fs.img_stride(), fs.H, fs.W, nbands, fs.colsum,
                               fs.sum, fs.sum_stride(), fs.sum4, fs.sum4_pitch(), fs.sum4_plane());
        else
            hipLaunchKernelGGL((k_integral4<512, true>), g2, dim3(512), 0, stream, fs.img, fs.img_stride(), fs.H, fs.W, nbands, fs.colsum,
                               fs.sum, fs.sum_stride(), fs.sum4, fs.sum4_pitch(), fs.sum4_plane());
        VO_HIP(hipGetLastError());
        return VO_OK;
    }
    const int cpt = (fs.W + 1023) / 1024;
#define LAUNCH_INT(C)                                                                                              \
    hipLaunchKernelGGL(k_integral<C>, g2, dim3(1024), 0, stream, fs.img, fs.img_stride(), fs.H, fs.W, nbands, \
                       fs.colsum, fs.sum, fs.sum_stride())
    if (cpt <= 1)
        LAUNCH_INT(1);
    else if (cpt <= 2)
        LAUNCH_INT(2);
    else if (cpt <= 4)
        LAUNCH_INT(4);
    else if (cpt <= 8)
        LAUNCH_INT(8);
    else
        return VO_ERR_UNSUPPORTED;
#undef LAUNCH_INT
    VO_HIP(hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------------------------------------
// a2 (inspection): one layer's det/trace, materialised.  Border samples are written as 0.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hessian_layer(const int32_t* __restrict__ sum, int H, int W, int layer_idx,
                                                       float* __restrict__ det, float* __restrict__ trace) {
    const LayerTab& L = c_tab.layers[layer_idx];
    const int lrows = H / L.step, lcols = W / L.step;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= lrows || j >= lcols) return;
    float dv = 0.f, tv = 0.f;
    if (L.size <= H && L.size <= W) {
        const int samples_i = 1 + (H - L.size) / L.step, samples_j = 1 + (W - L.size) / L.step;
        const int oi = i - L.margin, oj = j - L.margin;
        if (oi >= 0 && oj >= 0 && oi < samples_i && oj < samples_j) {
            float dx, dy, dxy;
            hessian_boxes(L, sum + (size_t)(oi * L.step) * (W + 1) + oj * L.step, W + 1, dx, dy, dxy);
            dv = hessian_det(dx, dy, dxy);
            tv = dx + dy;
        }
    }
    det[(size_t)i * lcols + j] = dv;
    trace[(size_t)i * lcols + j] = tv;
}

int vo_enqueue_hessian_layer(vo_ctx* ctx, FrameSet& fs, int frame, int octave, int layer, float* d_det,
                             float* d_trace) {
    const int step = 1 << octave;
    const int lrows = fs.H / step, lcols = fs.W / step;
    if (lrows <= 0 || lcols <= 0) return VO_OK;
    dim3 g((lcols + 63) / 64, (lrows + 3) / 4);
    hipLaunchKernelGGL(k_hessian_layer, g, dim3(256), 0, ctx->stream, fs.sum + (size_t)frame * fs.sum_stride(), fs.H,
                       fs.W, octave * kLayersPerOctave + layer, d_det, d_trace);
    VO_HIP(hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------------------------------------
// a2 + a3 fused: Hessian of the 5 layers of one octave for a 32x32 sample tile (+1 halo) into
// LDS, then NMS / interpolation / ballot compaction.  One launch covers all octaves and frames.
// ------------------------------------------------------------------------------------------------
constexpr int kTX = 32, kTY = 32, kHX = kTX + 2, kHY = kTY + 2, kHP = kHX + 1;

struct DetectGeom {
    int H, W;
    int tiles_x[kNumOctaves], tiles_y[kNumOctaves];
    int tile_base[kNumOctaves + 1];
};

// SURFInvoker deletes a keypoint when the gradient wavelet does not fit the integral image or
// when none of the 109 orientation samples lies inside it; the survivors keep their
// order, so dropping such keypoints here is equivalent to A.1 step 5.
// (upright - filter mode 2 - keeps only the first test: without an orientation search no sample has to fit)
__device__ bool kp_describable(float cx, float cy, float size, int H, int W, bool upright) {
    const float s = size * 1.2f / 9.0f;
    const int g = 2 * d_cvRoundF(2 * s);
    if (H + 1 < g || W + 1 < g) return false;
    if (upright) return true;
    const float half = (float)(g - 1) / 2;
    for (int kk = 0; kk < kOriSamples; kk++) {
        int x = d_cvRoundF(cx + c_tab.aptx[kk] * s - half);
        int y = d_cvRoundF(cy + c_tab.apty[kk] * s - half);
        if (!(y < 0 || y >= H + 1 - g || x < 0 || x >= W + 1 - g)) return true;
    }
    return false;
}

// a3: non-maximum suppression + interpolation of one tile whose five det planes are in LDS (plane p, row y, column x at
// s_det[p * PLANE + y * DP + x]; the interior sample (yy, xx) sits at row yy + 1, column xx + 1).
// Two passes: every interior sample of layers 1..3 is tested against the threshold and its 8 neighbours of the same
// layer, survivors (a few per cent) are compacted into an LDS queue; the queue is then walked with full wavefronts for
// the 18 neighbours of the adjacent layers, the interpolation and the append.  Keypoints are appended in
// arbitrary order (a4 sorts them by a total order).
constexpr int kNmsQueue = 768;  // entries: strict 3 x 3 maxima cannot be neighbours, so at most 16 x 16 per layer of a 32 x 32 tile, 3 layers

template <int TI, int DP, int PLANE>
__device__ __forceinline__ void nms_tile(const float* s_det, int* s_queue, int o, int ty, int tx, int H, int W, float thr,
                                         const int32_t* __restrict__ sum, vo_keypoint* __restrict__ kps, int cap,
                                         FrameCounters* __restrict__ cnt_frame, int filter_invalid,
                                         float* __restrict__ dbg_det) {
    const int tid = threadIdx.x, lane = tid & 63, NT = blockDim.x;
    const int step = 1 << o, sw = W + 1;
    const int lrows = H / step, lcols = W / step;
    if (dbg_det) {
        // inspection (vo_hessian_product_det): the det planes exactly as the NMS below reads them, interior of this tile, all
        // five layers of the octave; planes packed layer after layer, [H >> o][W >> o] each.  Null in every product launch.
        size_t off = 0;
        for (int q = 0; q < o; q++) off += (size_t)kLayersPerOctave * (H >> q) * (W >> q);
        for (int s = tid; s < TI * TI; s += NT) {
            const int yy = s / TI, xx = s - yy * TI;
            const int i = ty * TI + yy, j = tx * TI + xx;
            if (i < lrows && j < lcols)
                for (int l = 0; l < kLayersPerOctave; l++)
                    dbg_det[off + ((size_t)l * lrows + i) * lcols + j] = s_det[l * PLANE + (yy + 1) * DP + (xx + 1)];
        }
    }
    if (tid == 0) s_queue[0] = 0;
    __syncthreads();
    // (sample loop outside, layers inside: a sample's tile coordinates and frame position are computed once for its three
    // layers; the order in which candidates reach the queue is irrelevant)
    int margin_l[3];
#pragma unroll
    for (int l = 1; l <= 3; l++) margin_l[l - 1] = (c_tab.layers[o * kLayersPerOctave + l + 1].size / 2) / step + 1;
    // tiles away from the frame border (all but the outermost ring): every sample is inside every layer's margin, no per-sample
    // bounds tests
    const int mmax = margin_l[2];
    const bool inner = ty * TI >= mmax && (ty + 1) * TI <= lrows - mmax && tx * TI >= mmax && (tx + 1) * TI <= lcols - mmax;
    auto pass1 = [&](auto inner_c) {
        constexpr bool INNER = decltype(inner_c)::value;
        for (int s0 = 0; s0 < TI * TI; s0 += NT) {
            const int s = s0 + tid;
            const int yy = s / TI, xx = s - yy * TI;
            const int i = ty * TI + yy, j = tx * TI + xx;
            const float* p0 = s_det + (yy + 1) * DP + (xx + 1);
#pragma unroll
            for (int l = 1; l <= 3; l++) {
                const int margin = margin_l[l - 1];
                bool cand = false;
                if (s < TI * TI && (INNER || (i >= margin && i < lrows - margin && j >= margin && j < lcols - margin))) {
                    const float* p = p0 + l * PLANE;
                    const float v = p[0];
                    if (v > thr)
                        cand = v > p[-DP - 1] && v > p[-DP] && v > p[-DP + 1] && v > p[-1] && v > p[1] && v > p[DP - 1] && v > p[DP] &&
                               v > p[DP + 1];
                }
                const unsigned long long m = __ballot(cand);
                if (m) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_queue[0], __popcll(m));
                    base = __shfl(base, 0, 64);
                    const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                    if (cand && pos < kNmsQueue) s_queue[1 + pos] = (l << 20) | (yy << 10) | xx;
                }
            }
        }
    };
    if (inner)
        pass1(std::true_type{});
    else
        pass1(std::false_type{});
    __syncthreads();
    const int nq = min(s_queue[0], kNmsQueue);
    for (int q0 = 0; q0 < nq; q0 += NT) {
        const int q = q0 + tid;
        bool emit = false;
        vo_keypoint kpt;
        if (q < nq) {
            const int code = s_queue[1 + q];
            const int l = code >> 20, yy = (code >> 10) & 1023, xx = code & 1023;
            const int i = ty * TI + yy, j = tx * TI + xx;
            const LayerTab& L = c_tab.layers[o * kLayersPerOctave + l];
            const int size = L.size;
            const int size_prev = c_tab.layers[o * kLayersPerOctave + l - 1].size;
            float N9[3][9];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int t = 0; t < 9; t++) N9[a][t] = s_det[(l - 1 + a) * PLANE + (yy + t / 3) * DP + (xx + t % 3)];
            const float val0 = N9[1][4];
            bool is_max = true;
#pragma unroll
            for (int a = 0; a < 3; a += 2)
#pragma unroll
                for (int t = 0; t < 9; t++) is_max = is_max && (val0 > N9[a][t]);
            if (is_max) {
                const int sum_i = step * (i - (size / 2) / step);
                const int sum_j = step * (j - (size / 2) / step);
                const float center_i = sum_i + (size - 1) * 0.5f;
                const float center_j = sum_j + (size - 1) * 0.5f;
                // interpolateKeypoint, Cramer's rule in float
                const float b0 = -(N9[1][5] - N9[1][3]) / 2;
                const float b1 = -(N9[1][7] - N9[1][1]) / 2;
                const float b2 = -(N9[2][4] - N9[0][4]) / 2;
                const float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
                const float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
                const float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
                const float a10 = a01;
                const float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
                const float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
                const float a20 = a02, a21 = a12;
                const float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
                float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
                float x0 = 0, x1 = 0, x2 = 0;
                if (d != 0) {
                    d = 1 / d;
                    x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
                    x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
                    x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
                }
                const bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
                if (ok) {
                    const int ds = size - size_prev;
                    kpt.x = center_j + x0 * step;
                    kpt.y = center_i + x1 * step;
                    kpt.size = (float)d_cvRoundF((float)size + x2 * ds);
                    kpt.angle = -1.f;
                    kpt.response = val0;
                    kpt.octave = o;
                    // trace at the same sample (the product path never materialises it)
                    float dx, dy, dxy;
                    hessian_boxes(L, sum + (size_t)((i - L.margin) * step) * sw + (j - L.margin) * step, sw, dx, dy, dxy);
                    const float tr = dx + dy;
                    kpt.class_id = (tr > 0) - (tr < 0);
                    emit = !filter_invalid || kp_describable(kpt.x, kpt.y, kpt.size, H, W, filter_invalid == 2);
                }
            }
        }
        // wavefront-ballot compaction: one atomic per wave
        const unsigned long long m = __ballot(emit);
        if (m) {
            int base = 0;
            if (lane == 0) base = atomicAdd(&cnt_frame->n_raw, __popcll(m));
            base = __shfl(base, 0, 64);
            if (emit) {
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (pos < cap) kps[pos] = kpt;
            }
        }
    }
}

constexpr int kGlobalTileDet = kLayersPerOctave * kHY * kHP * 4;
constexpr int kGlobalTileLds = kGlobalTileDet + (kNmsQueue + 1) * 4;

__device__ __forceinline__ void hessian_nms_global_tile(unsigned char* smem, int b, int frame,
                                                        const int32_t* __restrict__ sum_base, size_t sum_stride,
                                                        const int32_t* __restrict__ sum4_base, int pw, size_t ps,
                                                        const DetectGeom& g, float thr, vo_keypoint* __restrict__ kp_base,
                                                        int cap, FrameCounters* __restrict__ cnt, int filter_invalid,
                                                        float* __restrict__ dbg_det) {
    float (*s_det)[kHY][kHP] = reinterpret_cast<float (*)[kHY][kHP]>(smem);
    int o = 0;
#pragma unroll
    for (int q = 1; q < kNumOctaves; q++)
        if (b >= g.tile_base[q]) o = q;
    b -= g.tile_base[o];
    const int ty = b / g.tiles_x[o], tx = b - ty * g.tiles_x[o];
    const int H = g.H, W = g.W;
    const int step = 1 << o;
    (void)0;
    const int32_t* __restrict__ sum = sum_base + (size_t)frame * sum_stride;
    const int32_t* __restrict__ planes = sum4_base + (size_t)frame * 4 * ps;
    const int tid = threadIdx.x;
    const int i0 = ty * kTY - 1, j0 = tx * kTX - 1;

    for (int l = 0; l < kLayersPerOctave; l++) {
        const LayerTab& L = c_tab.layers[o * kLayersPerOctave + l];
        const bool fits = (L.size <= H && L.size <= W);
        const int samples_i = fits ? 1 + (H - L.size) / step : 0;
        const int samples_j = fits ? 1 + (W - L.size) / step : 0;
        for (int s = tid; s < kHY * kHX; s += (int)blockDim.x) {
            const int yy = s / kHX, xx = s - yy * kHX;
            const int oi = i0 + yy - L.margin, oj = j0 + xx - L.margin;
            float dv = 0.f;
            if (oi >= 0 && oj >= 0 && oi < samples_i && oj < samples_j) {
                float dx, dy, dxy;
                // (step is 4 or 8 here: the origin column oj * step is a multiple of 4)
                hessian_boxes_planar(L, PlanarTap{planes, 4u * ((uint32_t)(oi * step) * (uint32_t)pw + (uint32_t)((oj * step) >> 2)), pw, ps}, dx, dy, dxy);
                dv = hessian_det(dx, dy, dxy);
            }
            s_det[l][yy][xx] = dv;
        }
    }
    __syncthreads();

    nms_tile<kTX, kHP, kHY * kHP>(&s_det[0][0][0], reinterpret_cast<int*>(smem + kGlobalTileDet), o, ty, tx, H, W, thr, sum,
                                  kp_base + (size_t)frame * cap, cap, cnt + frame, filter_invalid, frame == 0 ? dbg_det : nullptr);
}

// ------------------------------------------------------------------------------------------------
// a2 + a3, octaves 0 and 1: the integral-image patch under a T x T sample tile (all five layers) is staged in
// LDS with coalesced loads, and the 32 box taps of every sample are LDS reads.  Straight from L1/L2 the taps
// run at the texture addresser's gather rate (~4 lanes/clk/CU, tools/microbench); the patch is read once.
// Octave O samples every 2^O-th pixel: columns are stored de-interleaved by (col mod 2^O) so that the lanes
// of a tile row read consecutive LDS words.
// ------------------------------------------------------------------------------------------------
template <int O, int PP, int HALF>
struct LdsTap {
    const __attribute__((address_space(3))) int32_t* base;
    __device__ __forceinline__ uint32_t operator()(int y, int x) const {
        return (uint32_t)base[y * PP + (x & ((1 << O) - 1)) * HALF + (x >> O)];
    }
};

template <int T, int O>
struct LdsTileCfg {
    static constexpr int TI = T - 2;                       // interior samples per side (the rim is the NMS halo)
    static constexpr int step = 1 << O;
    static constexpr int kSizeMax = (9 + 6 * (kLayersPerOctave - 1)) << O;
    static constexpr int kMarginMax = (kSizeMax / 2) / step;
    static constexpr int PE = (T - 1) * step + kSizeMax + 1;  // patch extent in integral-image pixels
    static constexpr int HALF = O == 0 ? 0 : (((PE + step - 1) / step + 1) | 1);  // words per column phase: odd, and small enough for three tiles per CU (3 x 53.3 KB)
    static constexpr int PP = O == 0 ? PE : step * HALF;
    static constexpr int DP = T + 1;
    static constexpr int kSumBytes = PE * PP * 4;
    // the det planes (and the NMS queue behind them) take the patch's place once every tap has been read: LDS per tile is the
    // larger of the two, not their sum (octave 1: 54 KB instead of 66 KB; octave 0: 25 KB instead of 38 KB)
    static constexpr int kDetBytes = kLayersPerOctave * T * DP * 4;
    static constexpr int kOverlayBytes = kDetBytes + (kNmsQueue + 1) * 4;
    static constexpr int kBytes = kSumBytes > kOverlayBytes ? kSumBytes : kOverlayBytes;
};

// Sample s of a tile -> (row, column) for the det phase.  A tap is one ds_read_b32 per lane: the LDS serves lanes 0-31 and 32-63
// in one cycle each when the 32 words lie in 32 different banks (word address mod 32).  Octave 0: a tile row is 32 consecutive
// words - row-major order is conflict-free.  Octave 1 (24-sample rows, row pitch 2 * PP = 236 = 12 mod 32): row-major order puts
// lanes 24-31 (the next row) on banks 12-19 again, every tap took two cycles per group (SQ_LDS_BANK_CONFLICT = 47 % of
// SQ_LDS_IDX_ACTIVE); blocks of 8 rows x 4 columns per 32 lanes land on bank offsets 0,12,24,4,16,28,8,20 (+0..3): all 32 distinct.
template <int T, int O>
struct DetOrder {
    static constexpr bool kBlocked = O == 1;
    static __device__ __forceinline__ void pos(int s, int& yy, int& xx) {
        if constexpr (kBlocked) {
            static_assert(O == 0 || (T % 8 == 0 && (2 * LdsTileCfg<T, O>::PP) % 32 == 12), "8 x 4 blocks are conflict-free for this pitch only");
            const int g = s >> 5, l = s & 31;
            constexpr int kColBlocks = T / 4;
            const int rb = g / kColBlocks, cb = g - rb * kColBlocks;
            yy = rb * 8 + (l >> 2);
            xx = cb * 4 + (l & 3);
        } else {
            yy = s / T;
            xx = s - yy * T;
        }
    }
};

template <int T, int O, int NTH>
__device__ __forceinline__ void hessian_nms_lds_tile(unsigned char* smem, int b, int frame,
                                                     const int32_t* __restrict__ sum_base, size_t sum_stride, int H, int W,
                                                     int tiles_x, float thr, vo_keypoint* __restrict__ kp_base, int cap,
                                                     FrameCounters* __restrict__ cnt, int filter_invalid,
                                                     float* __restrict__ dbg_det) {
    using C = LdsTileCfg<T, O>;
    constexpr int TI = C::TI, step = C::step, kMarginMax = C::kMarginMax, PE = C::PE, HALF = C::HALF, PP = C::PP, DP = C::DP;
    int32_t* s_sum = reinterpret_cast<int32_t*>(smem);
    float (*s_det)[T][DP] = reinterpret_cast<float (*)[T][DP]>(smem);  // overlays the patch after the barrier below
    const int o = O;
    const int ty = b / tiles_x, tx = b - ty * tiles_x;
    const int sw = W + 1;
    (void)0;
    const int32_t* __restrict__ sum = sum_base + (size_t)frame * sum_stride;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int i0 = ty * TI - 1, j0 = tx * TI - 1;
    const int prow0 = (i0 - kMarginMax) * step, pcol0 = (j0 - kMarginMax) * step;

    {
        // The loads write LDS directly (global_load_lds_dword: no VGPR destination, no ds_write pass), so a wavefront has ALL its
        // rows in flight at once instead of four at a time (octave-0 launch 5.17 -> 4.80 ms).  The destination of such a load is
        // wave-uniform base + 4 * lane: octave 0's patch is row-major (64 consecutive columns -> 64 consecutive words); octave 1
        // stores a row's columns by parity, so a load takes every other column (lane k reads column 2k + phase).  Words outside
        // the frame stay as they are: no kept sample reads them.
        static_assert(HALF == 0 || HALF <= 64, "one load per (row, column phase)");
        constexpr int KC = O == 0 ? (PE + 63) / 64 : step;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), ln = tid & 63, nwv = NT >> 6;
        for (int pr = wv; pr < PE; pr += nwv) {
            const int gr = prow0 + pr;
            if (gr < 0 || gr > H) continue;  // (uniform over the wavefront)
#pragma unroll
            for (int k = 0; k < KC; k++) {
                const int pc = O == 0 ? ln + 64 * k : ln * step + k, gc = pcol0 + pc;
                const int dst = O == 0 ? pr * PP + 64 * k : pr * PP + k * HALF;
                if (pc < PE && gc >= 0 && gc <= W)
                    __builtin_amdgcn_global_load_lds(sum + (size_t)gr * sw + gc, (__attribute__((address_space(3))) void*)(s_sum + dst), 4, 0, 0);
            }
        }
    }
    __syncthreads();
    // det of sample s of layer l; the box corners are compile-time constants: every tap is a ds_read with an immediate offset
    auto det_at = [&](auto lc, int s) -> float {
        constexpr int l = decltype(lc)::value;
        constexpr int lsize = (9 + 6 * l) << O;
        constexpr BoxGeom G = box_geom(lsize);
        constexpr BoxWeights WG = box_weights(lsize);
        constexpr int lmargin = (lsize / 2) / step;
        const bool fits = (lsize <= H && lsize <= W);
        const int samples_i = fits ? 1 + (H - lsize) / step : 0;
        const int samples_j = fits ? 1 + (W - lsize) / step : 0;
        constexpr int dm = kMarginMax - lmargin;
        int yy, xx;
        DetOrder<T, O>::pos(s, yy, xx);
        const int oi = i0 + yy - lmargin, oj = j0 + xx - lmargin;
        float dv = 0.f;
        if (oi >= 0 && oj >= 0 && oi < samples_i && oj < samples_j) {
            float dx, dy, dxy;
            const LdsTap<O, PP, HALF> tap{(const __attribute__((address_space(3))) int32_t*)s_sum + (yy + dm) * step * PP + (xx + dm)};
            hessian_boxes_t(G, WG.w, tap, dx, dy, dxy);
            dv = hessian_det(dx, dy, dxy);
        }
        return dv;
    };
    // whole rounds of NTH samples, layer by layer; the results stay in registers until the patch is dead ...
    constexpr int kRounds = T * T / NTH, kFull = kRounds * NTH, kRem = T * T - kFull;
    float dvr[kRounds > 0 ? kRounds : 1][kLayersPerOctave];
#pragma unroll
    for (int it = 0; it < kRounds; it++) dvr[it][0] = det_at(std::integral_constant<int, 0>{}, tid + it * NTH);
#pragma unroll
    for (int it = 0; it < kRounds; it++) dvr[it][1] = det_at(std::integral_constant<int, 1>{}, tid + it * NTH);
#pragma unroll
    for (int it = 0; it < kRounds; it++) dvr[it][2] = det_at(std::integral_constant<int, 2>{}, tid + it * NTH);
#pragma unroll
    for (int it = 0; it < kRounds; it++) dvr[it][3] = det_at(std::integral_constant<int, 3>{}, tid + it * NTH);
#pragma unroll
    for (int it = 0; it < kRounds; it++) dvr[it][4] = det_at(std::integral_constant<int, 4>{}, tid + it * NTH);
    // ... and the remainder (T * T = 576 = 512 + 64 for the octave-1 tile) with ONE LAYER PER WAVEFRONT, all five at once,
    // instead of five more rounds in which a single wavefront of the eight has work
    float dv_rem = 0.f;
    const int wv_rem = tid >> 6, s_rem = kFull + (tid & 63);
    const bool has_rem = kRem > 0 && (tid & 63) < kRem && wv_rem < kLayersPerOctave;
    if (kRem > 0) {
        static_assert(kRem == 0 || (kRem <= 64 && NTH >= 64 * kLayersPerOctave), "remainder mapping: one wavefront per layer");
        if (has_rem) {
            switch (wv_rem) {
                case 0: dv_rem = det_at(std::integral_constant<int, 0>{}, s_rem); break;
                case 1: dv_rem = det_at(std::integral_constant<int, 1>{}, s_rem); break;
                case 2: dv_rem = det_at(std::integral_constant<int, 2>{}, s_rem); break;
                case 3: dv_rem = det_at(std::integral_constant<int, 3>{}, s_rem); break;
                default: dv_rem = det_at(std::integral_constant<int, 4>{}, s_rem); break;
            }
        }
    }
    __syncthreads();  // every tap has been read: the patch is dead, the det planes take its place
#pragma unroll
    for (int it = 0; it < kRounds; it++) {
        int yy, xx;
        DetOrder<T, O>::pos(tid + it * NTH, yy, xx);
#pragma unroll
        for (int l = 0; l < kLayersPerOctave; l++) s_det[l][yy][xx] = dvr[it][l];
    }
    if (has_rem) {
        int yy, xx;
        DetOrder<T, O>::pos(s_rem, yy, xx);
        s_det[wv_rem][yy][xx] = dv_rem;
    }
    __syncthreads();

    // the candidate queue sits behind the det planes
    nms_tile<TI, DP, T * DP>(&s_det[0][0][0], reinterpret_cast<int*>(smem + C::kDetBytes), o, ty, tx, H, W, thr, sum, kp_base + (size_t)frame * cap,
                             cap, cnt + frame, filter_invalid, frame == 0 ? dbg_det : nullptr);
}

// octave-0 tiles (LDS patch) and the few tiles of octaves >= OG (taps from L1/L2) in ONE grid, so that the long,
// latency-bound workgroups of the coarse octaves overlap the bulk of the work
template <int T0, int NTH>
__global__ __launch_bounds__(NTH) void k_hessian_nms_fused(const int32_t* __restrict__ sum_base, size_t sum_stride,
                                                           const int32_t* __restrict__ sum4_base, int pw, size_t ps, int H, int W,
                                                           int tiles_x0, int n_tiles0, DetectGeom g_rest, float thr,
                                                           vo_keypoint* __restrict__ kp_base, int cap,
                                                           FrameCounters* __restrict__ cnt, int filter_invalid,
                                                           float* __restrict__ dbg_det) {
    constexpr int kBytes = LdsTileCfg<T0, 0>::kBytes > kGlobalTileLds ? LdsTileCfg<T0, 0>::kBytes : kGlobalTileLds;
    __shared__ __attribute__((aligned(16))) unsigned char smem[kBytes];
    const int b = blockIdx.x, frame = blockIdx.y;
    // coarse tiles first in the grid: they are the long ones
    const int n_rest = g_rest.tile_base[kNumOctaves];
    if (b < n_rest)
        hessian_nms_global_tile(smem, b, frame, sum_base, sum_stride, sum4_base, pw, ps, g_rest, thr, kp_base, cap, cnt, filter_invalid, dbg_det);
    else
        hessian_nms_lds_tile<T0, 0, NTH>(smem, b - n_rest, frame, sum_base, sum_stride, H, W, tiles_x0, thr, kp_base, cap, cnt,
                                    filter_invalid, dbg_det);
}

template <int T, int O, int NTH>
__global__ __launch_bounds__(NTH) void k_hessian_nms_lds(const int32_t* __restrict__ sum_base, size_t sum_stride, int H, int W,
                                                         int tiles_x, float thr, vo_keypoint* __restrict__ kp_base, int cap,
                                                         FrameCounters* __restrict__ cnt, int filter_invalid,
                                                         float* __restrict__ dbg_det) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[LdsTileCfg<T, O>::kBytes];
    hessian_nms_lds_tile<T, O, NTH>(smem, blockIdx.x, blockIdx.y, sum_base, sum_stride, H, W, tiles_x, thr, kp_base, cap, cnt,
                               filter_invalid, dbg_det);
}

// ------------------------------------------------------------------------------------------------
// a2 + a3 for n_octave_layers != 3.  The reference fixes three layers per octave (camera_utility.hpp:36, a mutable global), and the
// kernels above are built for exactly that pyramid (five planes per tile in LDS, box weights as literals).  Any other layer count
// takes this plain form of calcLayerDetAndTrace / findMaximaInLayer: the det planes of one octave are
// materialised for a group of frames, one launch per layer, and one launch per middle layer does the 3 x 3 x 3 test, the
// interpolation and the append.  Same arithmetic, same keypoints as the oracle's; correctness path, no LDS form.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hessian_plane_g(const int32_t* __restrict__ sum_base, size_t sum_stride, int frame0, int H,
                                                         int W, LayerTab L, float* __restrict__ det, size_t frame_stride) {
    const int lrows = H / L.step, lcols = W / L.step;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= lrows || j >= lcols) return;
    const int32_t* __restrict__ sum = sum_base + (size_t)(frame0 + blockIdx.z) * sum_stride;
    float dv = 0.f;
    if (L.size <= H && L.size <= W) {
        const int samples_i = 1 + (H - L.size) / L.step, samples_j = 1 + (W - L.size) / L.step;
        const int oi = i - L.margin, oj = j - L.margin;
        if (oi >= 0 && oj >= 0 && oi < samples_i && oj < samples_j) {
            float dx, dy, dxy;
            hessian_boxes(L, sum + (size_t)(oi * L.step) * (W + 1) + oj * L.step, W + 1, dx, dy, dxy);
            dv = hessian_det(dx, dy, dxy);
        }
    }
    det[(size_t)blockIdx.z * frame_stride + (size_t)i * lcols + j] = dv;
}

// det: the middle layer's plane of frame frame0 + blockIdx.z; the layers below and above lie one plane (lrows * lcols) before / after
__global__ __launch_bounds__(256) void k_nms_g(const float* __restrict__ det, size_t frame_stride, LayerTab L, int size_prev,
                                               int size_next, int o, int H, int W, float thr, const int32_t* __restrict__ sum_base,
                                               size_t sum_stride, int frame0, vo_keypoint* __restrict__ kp_base, int cap,
                                               FrameCounters* __restrict__ cnt, int filter_invalid) {
    const int step = L.step, sw = W + 1;
    const int lrows = H / step, lcols = W / step;
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 64 + lane;
    const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = frame0 + blockIdx.z;
    const int margin = (size_next / 2) / step + 1;
    const size_t plane = (size_t)lrows * lcols;
    bool emit = false;
    vo_keypoint kpt;
    if (i >= margin && i < lrows - margin && j >= margin && j < lcols - margin) {
        const float* __restrict__ c = det + (size_t)blockIdx.z * frame_stride + (size_t)i * lcols + j;
        const float val0 = c[0];
        if (val0 > thr) {
            float N9[3][9];
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int t = 0; t < 9; t++) N9[a][t] = c[((ptrdiff_t)a - 1) * (ptrdiff_t)plane + (t / 3 - 1) * lcols + (t % 3 - 1)];
            bool is_max = true;
#pragma unroll
            for (int a = 0; a < 3; a++)
#pragma unroll
                for (int t = 0; t < 9; t++)
                    if (!(a == 1 && t == 4)) is_max = is_max && (val0 > N9[a][t]);
            if (is_max) {
                const int size = L.size;
                const int sum_i = step * (i - (size / 2) / step);
                const int sum_j = step * (j - (size / 2) / step);
                const float center_i = sum_i + (size - 1) * 0.5f;
                const float center_j = sum_j + (size - 1) * 0.5f;
                // interpolateKeypoint, Cramer's rule in float
                const float b0 = -(N9[1][5] - N9[1][3]) / 2;
                const float b1 = -(N9[1][7] - N9[1][1]) / 2;
                const float b2 = -(N9[2][4] - N9[0][4]) / 2;
                const float a00 = N9[1][3] - 2 * N9[1][4] + N9[1][5];
                const float a01 = (N9[1][8] - N9[1][6] - N9[1][2] + N9[1][0]) / 4;
                const float a02 = (N9[2][5] - N9[2][3] - N9[0][5] + N9[0][3]) / 4;
                const float a10 = a01;
                const float a11 = N9[1][1] - 2 * N9[1][4] + N9[1][7];
                const float a12 = (N9[2][7] - N9[2][1] - N9[0][7] + N9[0][1]) / 4;
                const float a20 = a02, a21 = a12;
                const float a22 = N9[0][4] - 2 * N9[1][4] + N9[2][4];
                float d = a00 * (a11 * a22 - a21 * a12) - a01 * (a10 * a22 - a20 * a12) + a02 * (a10 * a21 - a20 * a11);
                float x0 = 0, x1 = 0, x2 = 0;
                if (d != 0) {
                    d = 1 / d;
                    x0 = d * (b0 * (a11 * a22 - a12 * a21) - a01 * (b1 * a22 - a12 * b2) + a02 * (b1 * a21 - a11 * b2));
                    x1 = d * (a00 * (b1 * a22 - a12 * b2) - b0 * (a10 * a22 - a12 * a20) + a02 * (a10 * b2 - b1 * a20));
                    x2 = d * (a00 * (a11 * b2 - b1 * a21) - a01 * (a10 * b2 - b1 * a20) + b0 * (a10 * a21 - a11 * a20));
                }
                const bool ok = (x0 != 0 || x1 != 0 || x2 != 0) && fabsf(x0) <= 1 && fabsf(x1) <= 1 && fabsf(x2) <= 1;
                if (ok) {
                    const int ds = size - size_prev;
                    kpt.x = center_j + x0 * step;
                    kpt.y = center_i + x1 * step;
                    kpt.size = (float)d_cvRoundF((float)size + x2 * ds);
                    kpt.angle = -1.f;
                    kpt.response = val0;
                    kpt.octave = o;
                    float dx, dy, dxy;  // the trace at the same sample (not materialised)
                    const int32_t* __restrict__ sum = sum_base + (size_t)frame * sum_stride;
                    hessian_boxes(L, sum + (size_t)((i - L.margin) * step) * sw + (j - L.margin) * step, sw, dx, dy, dxy);
                    const float tr = dx + dy;
                    kpt.class_id = (tr > 0) - (tr < 0);
                    emit = !filter_invalid || kp_describable(kpt.x, kpt.y, kpt.size, H, W, filter_invalid == 2);
                }
            }
        }
    }
    const unsigned long long m = __ballot(emit);
    if (m) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&cnt[frame].n_raw, __popcll(m));
        base = __shfl(base, 0, 64);
        if (emit) {
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) kp_base[(size_t)frame * cap + pos] = kpt;
        }
    }
}

static int enqueue_hessian_nms_generic(vo_ctx* ctx, FrameSet& fs, bool describe, int o_begin) {
    const int nl = ctx->prm.n_octave_layers, n_oct = ctx->prm.n_octaves;
    const int filter_mode = describe ? (ctx->prm.upright ? 2 : 1) : 0;
    // det planes of ONE octave for a group of frames; the buffer is bounded (2 GB) and grows with the context
    const size_t per_frame0 = (size_t)(nl + 2) * (fs.H >> o_begin) * (fs.W >> o_begin);  // floats, the first (largest) octave
    if (per_frame0 == 0) return VO_OK;
    const size_t budget = (size_t)1 << 29;  // floats
    const int group = (int)std::min<size_t>((size_t)fs.F, std::max<size_t>(1, budget / per_frame0));
    const size_t need = per_frame0 * group * sizeof(float);
    if (need > ctx->d_gen_bytes) {
        if (ctx->d_gen) VO_HIP(hipFree(ctx->d_gen));  // (synchronises: nothing still reads the old buffer)
        ctx->d_gen = nullptr;
        ctx->d_gen_bytes = 0;
        VO_HIP(hipMalloc(&ctx->d_gen, need));
        ctx->d_gen_bytes = need;
    }
    float* det = (float*)ctx->d_gen;
    StageScope sc(ctx, ST_HESSIAN_OCT0);
    for (int f0 = 0; f0 < fs.F; f0 += group) {
        const int nf = std::min(group, fs.F - f0);
        for (int o = o_begin; o < n_oct; o++) {
            const int step = 1 << o, lrows = fs.H / step, lcols = fs.W / step;
            if (lrows <= 0 || lcols <= 0) continue;
            const size_t plane = (size_t)lrows * lcols, frame_stride = plane * (nl + 2);
            const dim3 g((lcols + 63) / 64, (lrows + 3) / 4, nf);
            std::vector<LayerTab> tabs(nl + 2);
            for (int l = 0; l < nl + 2; l++) {
                tabs[l] = make_layer_tab((9 + 6 * l) << o, step);
                hipLaunchKernelGGL(k_hessian_plane_g, g, dim3(256), 0, ctx->stream, fs.sum, fs.sum_stride(), f0, fs.H, fs.W, tabs[l],
                                   det + (size_t)l * plane, frame_stride);
            }
            for (int l = 1; l <= nl; l++)
                hipLaunchKernelGGL(k_nms_g, g, dim3(256), 0, ctx->stream, det + (size_t)l * plane, frame_stride, tabs[l], tabs[l - 1].size,
                                   tabs[l + 1].size, o, fs.H, fs.W, ctx->prm.hessian_threshold, fs.sum, fs.sum_stride(), f0, fs.kp_raw,
                                   fs.cap, fs.cnt, filter_mode);
        }
    }
    VO_HIP(hipGetLastError());
    return VO_OK;
}

// ------------------------------------------------------------------------------------------------
// a4: KeypointGreater rank sort.  rank[i] = #{j : kp_j > kp_i}; the order is total, so ranks
// are a permutation and the scatter is deterministic whatever order NMS appended in.
// ------------------------------------------------------------------------------------------------
constexpr int kRankChunk = 1024;

__global__ __launch_bounds__(256) void k_count_clamp(FrameCounters* cnt, int cap, int F) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f < F) cnt[f].n_kp = min(cnt[f].n_raw, cap);
}

__global__ __launch_bounds__(256) void k_rank(const vo_keypoint* __restrict__ kp_base, int cap,
                                              const FrameCounters* __restrict__ cnt, int* __restrict__ rank_base) {
    __shared__ float s_r[kRankChunk];
    const int frame = blockIdx.z;
    const int n = min(cnt[frame].n_raw, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j0 = blockIdx.y * kRankChunk;
    if (blockIdx.x * 256 >= n || j0 >= n) return;
    const vo_keypoint* kps = kp_base + (size_t)frame * cap;
    const int jn = min(kRankChunk, n - j0);
    for (int q = threadIdx.x; q < jn; q += 256) s_r[q] = kps[j0 + q].response;
    __syncthreads();
    if (i >= n) return;
    const vo_keypoint me = kps[i];
    int c = 0;
    for (int q = 0; q < jn; q++) {
        const float r = s_r[q];
        c += (r > me.response);
        if (r == me.response) {  // rare: full KeypointGreater chain
            const vo_keypoint o = kps[j0 + q];
            bool gt;
            if (o.size != me.size)
                gt = o.size > me.size;
            else if (o.octave != me.octave)
                gt = o.octave > me.octave;
            else if (o.y != me.y)
                gt = o.y > me.y;
            else
                gt = o.x < me.x;
            c += gt;
        }
    }
    if (c) atomicAdd(&rank_base[(size_t)frame * cap + i], c);
}

__global__ __launch_bounds__(256) void k_scatter(const vo_keypoint* __restrict__ kp_base, int cap,
                                                 const FrameCounters* __restrict__ cnt, int* __restrict__ rank_base,
                                                 vo_keypoint* __restrict__ out_base) {
    const int frame = blockIdx.y;
    const int n = min(cnt[frame].n_raw, cap);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const size_t o = (size_t)frame * cap;
    const int r = rank_base[o + i];
    rank_base[o + i] = 0;  // leave the rank array zeroed for the next run
    if (r < n) out_base[o + r] = kp_base[o + i];
}

// a4, up to 16384 keypoints per frame: bitonic sort of (response, index) keys in LDS, one 1024-thread workgroup per
// frame.  The key order is KeypointGreater: the response decides (positive floats order like their bit patterns);
// equal responses (rare) run the rest of the chain - size, octave, y, x - on the records.  The order is total, so the
// network's result is the unique sorted sequence whatever order NMS appended in.
constexpr int kBitonicMax = 16384;

__device__ __forceinline__ bool kp_key_greater(unsigned long long a, unsigned long long b, const vo_keypoint* __restrict__ kps) {
    const unsigned ra = (unsigned)(a >> 32), rb = (unsigned)(b >> 32);
    if (ra != rb) return ra > rb;
    const unsigned ia = (unsigned)a, ib = (unsigned)b;
    if (ia == 0xffffffffu || ib == 0xffffffffu) return ib == 0xffffffffu && ia != 0xffffffffu;  // padding sorts last
    const vo_keypoint p = kps[ia], q = kps[ib];
    if (p.size != q.size) return p.size > q.size;
    if (p.octave != q.octave) return p.octave > q.octave;
    if (p.y != q.y) return p.y > q.y;
    return p.x < q.x;
}

__global__ __launch_bounds__(1024) void k_sort_bitonic(const vo_keypoint* __restrict__ kp_base, int cap,
                                                       const FrameCounters* __restrict__ cnt,
                                                       vo_keypoint* __restrict__ out_base) {
    extern __shared__ unsigned long long s_key[];
    const int frame = blockIdx.x, tid = threadIdx.x;
    const int n = min(cnt[frame].n_raw, cap);
    if (n <= 0) return;
    const vo_keypoint* kps = kp_base + (size_t)frame * cap;
    int np = 1024;  // at least one compare-exchange per thread pair keeps the loops simple
    while (np < n) np <<= 1;
    for (int i = tid; i < np; i += 1024) {
        unsigned long long key = 0xffffffffull;  // padding: lowest response, index -1
        if (i < n) {
            const unsigned bits = __float_as_uint(kps[i].response);
            const unsigned mono = bits ^ ((bits >> 31) ? 0xffffffffu : 0x80000000u);  // float order -> unsigned order
            key = ((unsigned long long)mono << 32) | (unsigned)i;
        }
        s_key[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= np; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np >> 1); t += 1024) {
                const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;
                const bool desc = (lo & k) == 0;  // first half of every k-block sorted descending: the whole is descending
                const unsigned long long a = s_key[lo], b = s_key[hi];
                const bool a_gt_b = kp_key_greater(a, b, kps);
                if (a_gt_b != desc) {
                    s_key[lo] = b;
                    s_key[hi] = a;
                }
            }
            __syncthreads();
        }
    }
    vo_keypoint* out = out_base + (size_t)frame * cap;
    for (int i = tid; i < n; i += 1024) out[i] = kps[(unsigned)s_key[i]];
}

// ------------------------------------------------------------------------------------------------
// a5 + a6: orientation and descriptor, one wavefront (= one 64-thread workgroup) per keypoint.
// ------------------------------------------------------------------------------------------------
constexpr int kSmallWin = 128;  // windows up to 128 px use the 11 KB LDS instance
constexpr int kMidWin = 320;
constexpr int kBigWin = 768;    // largest window: 21 * 1.2/9 * (216 + 48) = 739
constexpr int kOct5Win = 1536;  // n_octaves = 5 (with at most three layers): 21 * 1.2/9 * (432 + 96) = 1478 (sequential kernel only)
constexpr int kHugeWin = 1152;  // n_octave_layers up to kMaxOctaveLayers = 6: 21 * 1.2/9 * (360 + 48) = 1142 (sequential kernel only)

struct CellTab {  // one destination cell of the INTER_AREA decimation table
    int sx1, sx2;         // full-weight source range [sx1, sx2)
    int has_head, has_tail;
    float a_head, a_mid, a_tail;
    int pad;
};

__device__ __forceinline__ int sample_pixel(const uint8_t* __restrict__ img, int W, int ncols1, int nrows1, double px,
                                            double py) {
    const int ix = d_cvFloorD(px), iy = d_cvFloorD(py);
    if ((unsigned)ix < (unsigned)ncols1 && (unsigned)iy < (unsigned)nrows1) {
        const float a = (float)(px - ix), b = (float)(py - iy);
        const uint8_t* p = img + (size_t)iy * W + ix;
        const int p00 = p[0], p01 = p[1], p10 = p[W], p11 = p[W + 1];
        const float oma = 1.f - a, omb = 1.f - b;
        float v = (float)p00 * oma * omb;
        v = v + (float)p01 * a * omb;
        v = v + (float)p10 * oma * b;
        v = v + (float)p11 * a * b;
        return (int)(uint8_t)d_cvRoundF(v);
    }
    const int x = min(max(d_cvRoundD(px), 0), ncols1);
    const int y = min(max(d_cvRoundD(py), 0), nrows1);
    return img[(size_t)y * W + x];
}

__device__ __forceinline__ void wave_sync_lds() {  // LDS hand-off between lanes of one wavefront
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;

__device__ __forceinline__ int saturate_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

// Decimation table per SOURCE pixel column j (identical for every row of the window, and, applied to
// rows, for the vertical pass): the pixel adds S*w_cur to the open destination cell; if it is the cell's
// last pixel the cell is closed and the next one opens with S*w_next (the pixel's head share, or 0).
struct PixTab {
    float w_cur, w_next;
    int end;
};

// orientation (A.5) of one keypoint by one wavefront; writes kpp->angle and *s_dir (lane 0)
__device__ __forceinline__ void describe_orientation(const int32_t* __restrict__ sum, int sw, int H, int W, float cx, float cy,
                                                     float s, int lane, int tid, float2* s_XY, int* s_A,
                                                     vo_keypoint* kpp, float* s_dir_p) {
    const int g = 2 * d_cvRoundF(2 * s);
    int nangle = 0;
    if (!(H + 1 < g || W + 1 < g)) {
        const int xm = g / 2;  // cvRound((float)g/4 * 2)
        const float wxa = -1.f / ((float)xm * g), wxb = 1.f / ((float)(g - xm) * g);
        const float wya = 1.f / ((float)g * xm), wyb = -1.f / ((float)g * (g - xm));
        const float half = (float)(g - 1) / 2;
        for (int pass = 0; pass < 2; pass++) {
            const int kk = pass * 64 + lane;
            bool inb = false;
            float X = 0, Y = 0;
            int Ar = 0;
            if (kk < kOriSamples) {
                const int x = d_cvRoundF(cx + c_tab.aptx[kk] * s - half);
                const int y = d_cvRoundF(cy + c_tab.apty[kk] * s - half);
                inb = !(y < 0 || y >= H + 1 - g || x < 0 || x >= W + 1 - g);
                if (inb) {
                    const int32_t* p = sum + (size_t)y * sw + x;
                    const uint32_t t00 = p[0], t0m = p[xm], t0g = p[g];
                    const uint32_t tm0 = p[(size_t)xm * sw], tmg = p[(size_t)xm * sw + g];
                    const uint32_t tg0 = p[(size_t)g * sw], tgm = p[(size_t)g * sw + xm], tgg = p[(size_t)g * sw + g];
                    // dx: boxes x[0,xm] (w<0) and x[xm,g] (w>0), both y[0,g]
                    const int bxa = (int)(t00 + tgm - tg0 - t0m), bxb = (int)(t0m + tgg - tgm - t0g);
                    double d = (double)((float)bxa * wxa);
                    d += (double)((float)bxb * wxb);
                    const float vx = (float)d;
                    // dy: boxes y[0,xm] (w>0) and y[xm,g] (w<0), both x[0,g]
                    const int bya = (int)(t00 + tmg - tm0 - t0g), byb = (int)(tm0 + tgg - tg0 - tmg);
                    d = (double)((float)bya * wya);
                    d += (double)((float)byb * wyb);
                    const float vy = (float)d;
                    const float w = c_tab.aptw[kk];
                    X = vx * w;
                    Y = vy * w;
                    Ar = d_cvRoundF(d_fastAtan2(Y, X));
                }
            }
            const unsigned long long m = __ballot(inb);
            if (inb) {
                const int pos = nangle + __popcll(m & ((1ull << lane) - 1ull));
                s_XY[pos] = make_float2(X, Y);
                s_A[pos] = Ar;
            }
            nangle += __popcll(m);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    nangle = __builtin_amdgcn_readfirstlane(nangle);  // the same in every lane (ballot counts): a scalar trip count
    float best_mod = 0.f, best_x = 0.f, best_y = 0.f;
    int best_i = 1 << 30;
    {
        // 72 windows, one pass: lane L < 36 sums window L (centre i0 = 5L) and the opposite window L + 36.  One
        // angular distance serves both: |A - i0| < 30 or > 330 is window L, 150 < |A - i0| < 210 is window L + 36.
        // Samples outside a window add +0.f, which leaves a float sum bit-identical to skipping them.
        const int i0 = (lane < 36 ? lane : 0) * 5, i1 = i0 + 180;
        typedef float f32x2_t __attribute__((ext_vector_type(2)));
        f32x2_t acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f};
        auto window_step = [&](int j) {
            const int A = s_A[j];
            const float2 xyv = s_XY[j];  // one 64-bit LDS read straight into the register pair of the packed add
            const f32x2_t xy = {xyv.x, xyv.y};
            // |A - i0| - 30 in one instruction (A and i0 are non-negative: the unsigned sum of absolute differences)
            unsigned dm30;
            asm("v_sad_u32 %0, %1, %2, %3" : "=v"(dm30) : "v"(A), "v"(i0), "v"(-30));
            // lanes whose window holds the sample add it under EXEC (one packed add per window; skipping a lane is the
            // same as adding +0.f): the kernel is bound by VALU issue, and selects would cost two more instructions each;
            // the empty asm keeps each add a real branch under EXEC instead of two selects + add
            if (dm30 > 300u) {
                acc0 += xy;
                asm volatile("" : "+v"(acc0));
            }
            if (dm30 - 121u < 59u) {
                acc1 += xy;
                asm volatile("" : "+v"(acc1));
            }
        };
        int j = 0;
        for (; j + 4 <= nangle; j += 4) {  // unrolled by hand (the branches under EXEC keep the compiler from doing it)
            window_step(j);
            window_step(j + 1);
            window_step(j + 2);
            window_step(j + 3);
        }
        for (; j < nangle; j++) window_step(j);
        const float sx0 = acc0.x, sy0 = acc0.y, sx1 = acc1.x, sy1 = acc1.y;
        const float m0 = sx0 * sx0 + sy0 * sy0;
        if (lane < 36 && m0 > best_mod) {
            best_mod = m0;
            best_x = sx0;
            best_y = sy0;
            best_i = i0;
        }
        const float m1 = sx1 * sx1 + sy1 * sy1;
        if (lane < 36 && m1 > best_mod) {  // i1 > i0: strictly greater only
            best_mod = m1;
            best_x = sx1;
            best_y = sy1;
            best_i = i1;
        }
    }
    // first strict maximum over all windows: larger mod wins, ties go to the smaller i
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const float om = __shfl_xor(best_mod, d, 64), ox = __shfl_xor(best_x, d, 64),
                    oy = __shfl_xor(best_y, d, 64);
        const int oi = __shfl_xor(best_i, d, 64);
        if (om > best_mod || (om == best_mod && oi < best_i)) {
            best_mod = om;
            best_x = ox;
            best_y = oy;
            best_i = oi;
        }
    }
    if (!(best_mod > 0.f)) {
        best_x = 0.f;
        best_y = 0.f;
    }
    const float dd = d_fastAtan2(-best_y, best_x);
    if (tid == 0) {
        kpp->angle = dd;
        *s_dir_p = dd;
    }
}

// computeResizeAreaTab (A.6.1): one destination cell per thread (tid <= kPatch)
__device__ __forceinline__ void describe_cell_tab(int tid, int win, double scale, CellTab* s_cell) {
    const int dxc = tid;
    const double fsx1 = dxc * scale, fsx2 = fsx1 + scale;
    const double cellWidth = fmin(scale, (double)win - fsx1);
    int sx1 = d_cvCeilD(fsx1), sx2 = d_cvFloorD(fsx2);
    sx2 = min(sx2, win - 1);
    sx1 = min(sx1, sx2);
    CellTab c;
    c.sx1 = sx1;
    c.sx2 = sx2;
    c.has_head = (sx1 - fsx1 > 1e-3);
    c.a_head = (float)((sx1 - fsx1) / cellWidth);
    c.a_mid = (float)(1.0 / cellWidth);
    c.has_tail = (fsx2 - sx2 > 1e-3);
    c.a_tail = (float)(fmin(fmin(fsx2 - sx2, 1.), cellWidth) / cellWidth);
    c.pad = 0;
    s_cell[dxc] = c;
}

// per-source-pixel view of the same table (see PixTab)
__device__ __forceinline__ void describe_pix_tab(int tid, int NT, int win, double scale, const CellTab* s_cell,
                                                 PixTab* s_pix) {
    for (int j = tid; j < win; j += NT) {
        // the (at most two) table entries that read source pixel j, in destination order
        PixTab t;
        t.w_cur = 0.f;
        t.w_next = 0.f;
        t.end = 0;
        bool have = false;
        // cell c reads pixels [c*scale - 1, (c+1)*scale]: only cells floor(j/scale) - 1 .. + 2 can hold pixel j
        const int cfirst = max((int)((double)j / scale) - 1, 0), clast = min(cfirst + 3, kPatch);
#pragma unroll 1
        for (int c = cfirst; c <= clast; c++) {
            const CellTab ct = s_cell[c];
            float w = 0.f;
            bool hit = false;
            if (ct.has_head && j == ct.sx1 - 1) {
                w = ct.a_head;
                hit = true;
            } else if (j >= ct.sx1 && j < ct.sx2) {
                w = ct.a_mid;
                hit = true;
            } else if (ct.has_tail && j == ct.sx2) {
                w = ct.a_tail;
                hit = true;
            }
            if (hit) {
                if (!have) {
                    have = true;
                    t.w_cur = w;
                    const int last = ct.has_tail ? ct.sx2 : ct.sx2 - 1;
                    t.end = (j == last) ? 1 : 0;
                } else {
                    t.w_next = w;
                }
            }
        }
        s_pix[j] = t;
    }
}

// 21x21 patch -> Gaussian-weighted gradients -> 4x4 cells x (dx, dy, |dx|, |dy|) -> unit vector (A.6.2, A.6.3).
// Called by every thread of the workgroup (contains barriers); s_grad holds 2 * 400 floats.
__device__ __forceinline__ void describe_patch_to_vec(int tid, int NT, const uint8_t* s_patch, float* s_grad, float* s_vec,
                                                      float* __restrict__ desc_out) {
    const int lane = tid & 63;
    float* s_DX = s_grad;
    float* s_DY = s_grad + kPatch * kPatch;
    // (the Gaussian weight is indexed per lane, i.e. it comes through a global load: the next iteration's is requested before
    // this iteration's arithmetic, so that its latency is not waited for seven times per keypoint in the one-wavefront kernels)
    float dw_next = c_tab.DW[min(tid, kPatch * kPatch - 1)];
    for (int it = tid; it < kPatch * kPatch; it += NT) {
        const int i = it / kPatch, j = it - i * kPatch;
        const float dw = dw_next;
        dw_next = c_tab.DW[min(it + NT, kPatch * kPatch - 1)];
        const int p00 = s_patch[i * (kPatch + 1) + j], p01 = s_patch[i * (kPatch + 1) + j + 1];
        const int p10 = s_patch[(i + 1) * (kPatch + 1) + j], p11 = s_patch[(i + 1) * (kPatch + 1) + j + 1];
        s_DX[it] = (float)(p01 - p00 + p11 - p10) * dw;
        s_DY[it] = (float)(p10 - p00 + p11 - p01) * dw;
    }
    __syncthreads();
    if (tid < 64) {
        // lane = cell*4 + q : 25 sequential float adds per lane, the order of the reference loops.  q picks the plane
        // (dx / dy) and whether the sign bit is cleared (|dx|, |dy|): one LDS read, one AND, one add per element.
        const int cell = lane >> 2, q = lane & 3;
        const int ci = cell >> 2, cj = cell & 3;
        const float* plane = (q & 1) ? s_DY : s_DX;
        const uint32_t keep = q >= 2 ? 0x7fffffffu : 0xffffffffu;
        float acc = 0.f;
#pragma unroll
        for (int y = 0; y < 5; y++)
#pragma unroll
            for (int x = 0; x < 5; x++) acc += __uint_as_float(__float_as_uint(plane[(ci * 5 + y) * kPatch + cj * 5 + x]) & keep);
        // square_mag = sum over kk of (double)(v*v), added in index order by the reference.  Every term is a float; when the
        // exponents of the non-zero terms span at most 23 binades, every partial sum of them - in ANY order - is a multiple of
        // 2^(emin-150) below 2^(emax-120) and fits a double's 53 bits, so a tree sum gives the sequential result bit for bit.
        // Otherwise (a cell sum ~3000 times smaller than another) the lanes walk the 64 values in order.
        const float t = acc * acc;
        const int e = (__float_as_int(t) >> 23) & 0xff;
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        u16x2 span = {(unsigned short)(t != 0.f ? e : 0), (unsigned short)(t != 0.f ? 255 - e : 0)};  // max exponent | 255 - min exponent
        double square_mag = (double)t;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)__builtin_bit_cast(uint32_t, span), m, 64);
            span = __builtin_elementwise_max(span, __builtin_bit_cast(u16x2, o));
            square_mag += __shfl_xor(square_mag, m, 64);
        }
        if ((int)span.x - (255 - (int)span.y) > 23) {  // wave-uniform
            s_vec[lane] = acc;
            wave_sync_lds();
            square_mag = 0;
            for (int kk = 0; kk < 64; kk++) {
                const float v = s_vec[kk];
                square_mag += (double)(v * v);
            }
        }
        const float nscale = (float)(1. / (sqrt(square_mag) + (double)FLT_EPSILON));
        desc_out[lane] = acc * nscale;
    }
}

// The 128-float form of the same step (SURF's `extended` flag; false in the reference, camera_utility.hpp:37; not part of SURVEY
// Appendix A - restated from the published OpenCV source, parity unpinned like `upright`): per cell the sums of dx and |dx| are
// kept apart by the sign of dy (dy >= 0 first), those of dy and |dy| by the sign of dx.  Thread = cell * 8 + q (needs NT >= 128);
// a term of the other sign is skipped, as in the reference's branches.  square_mag is the reference's sequential double sum.
// s_vec holds 128 floats.
__device__ __forceinline__ void describe_patch_to_vec_ext(int tid, int NT, const uint8_t* s_patch, float* s_grad, float* s_vec,
                                                          float* __restrict__ desc_out) {
    float* s_DX = s_grad;
    float* s_DY = s_grad + kPatch * kPatch;
    for (int it = tid; it < kPatch * kPatch; it += NT) {
        const int i = it / kPatch, j = it - i * kPatch;
        const float dw = c_tab.DW[it];
        const int p00 = s_patch[i * (kPatch + 1) + j], p01 = s_patch[i * (kPatch + 1) + j + 1];
        const int p10 = s_patch[(i + 1) * (kPatch + 1) + j], p11 = s_patch[(i + 1) * (kPatch + 1) + j + 1];
        s_DX[it] = (float)(p01 - p00 + p11 - p10) * dw;
        s_DY[it] = (float)(p10 - p00 + p11 - p01) * dw;
    }
    __syncthreads();
    float acc = 0.f;
    if (tid < 128) {
        const int cell = tid >> 3, q = tid & 7;
        const int ci = cell >> 2, cj = cell & 3;
        const float* val = (q & 4) ? s_DY : s_DX;    // what is summed
        const float* cond = (q & 4) ? s_DX : s_DY;   // whose sign selects
        const bool want_nonneg = (q & 2) == 0, absolute = (q & 1) != 0;
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) {
                const int o = (ci * 5 + y) * kPatch + cj * 5 + x;
                const float v = val[o];
                if ((cond[o] >= 0.f) == want_nonneg) acc += absolute ? fabsf(v) : v;
            }
        s_vec[tid] = acc;
    }
    __syncthreads();
    if (tid < 128) {
        double square_mag = 0;
        for (int kk = 0; kk < 128; kk++) {
            const float v = s_vec[kk];
            square_mag += (double)(v * v);
        }
        const float nscale = (float)(1. / (sqrt(square_mag) + (double)FLT_EPSILON));
        desc_out[tid] = acc * nscale;
    }
}

// The reference's own walk, kept as the fallback for keypoints whose window coordinates are not exact in closed form
// (see k_describe_tiled): correctness over speed.  NT threads cooperate on one keypoint of list `cls`.
template <int WLO, int WHI, int NT>
__global__ __launch_bounds__(NT) void k_describe(const uint8_t* __restrict__ img_base, size_t img_stride,
                                                 const int32_t* __restrict__ sum_base, size_t sum_stride, int H, int W,
                                                 vo_keypoint* __restrict__ kp_base, float* __restrict__ desc_base,
                                                 int cap, const int* __restrict__ list, int* __restrict__ cls_cnt, int cls, int have_dir,
                                                 int extended) {
    __shared__ float2 s_XY[kOriSamples + 15];
    __shared__ int s_A[kOriSamples + 15];
    __shared__ CellTab s_cell[kPatch + 1];
    __shared__ PixTab s_pix[WHI];
    __shared__ float s_rowbuf[(WHI < 40 ? 40 : WHI) * (kPatch + 1)];  // float or int row accumulators; later DX/DY
    __shared__ uint8_t s_patch[(kPatch + 1) * (kPatch + 1) + 7];
    __shared__ float s_vec[128];
    __shared__ float s_dir;
    __shared__ int s_item;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int sw = W + 1;
    const int n_items = cls_cnt[cls];
    static_assert(NT >= 128, "describe_patch_to_vec_ext");
    // work queue: keypoints of this window class, handed out one per workgroup by an atomic ticket; every
    // workgroup leaves as soon as its ticket is past the end of the list
    for (;;) {
    __syncthreads();  // previous keypoint's LDS is no longer read
    if (tid == 0) s_item = atomicAdd(&cls_cnt[ticket_index(cls, 0)], 1);
    __syncthreads();
    const int item = s_item;
    if (item >= n_items) break;
    const int code = list[item];
    const int frame = code / cap, k = code - frame * cap;
    vo_keypoint* kpp = kp_base + (size_t)frame * cap + k;
    const float size = kpp->size;
    const float cx = kpp->x, cy = kpp->y;
    const float s = size * 1.2f / 9.0f;
    const int win = (int)((kPatch + 1) * s);
    const uint8_t* __restrict__ img = img_base + (size_t)frame * img_stride;
    const int32_t* __restrict__ sum = sum_base + (size_t)frame * sum_stride;

    // ---------------------------------------------------------------- orientation (A.5), first wavefront
    // (have_dir: k_orientation has run and the record holds the direction - the same function gave it; the integral image is
    // then not read here, so the NEXT detection's integral pass may already be overwriting it on the auxiliary stream)
    const bool upright = have_dir == 2;  // SURF's upright flag: the window is an axis-aligned rectangle of pixels (no bilinear walk)
    if (have_dir) {
        if (tid == 0) s_dir = kpp->angle;
    } else if (tid < 64) {
        describe_orientation(sum, sw, H, W, cx, cy, s, lane, tid, s_XY, s_A, kpp, &s_dir);
    }
    // ---------------------------------------------------------------- decimation tables (A.6.1)
    const double scale = (double)win / (kPatch + 1);
    const int iscale = d_cvRoundD(scale);
    const bool area_fast = fabs(scale - iscale) < DBL_EPSILON;
    if (!area_fast && tid <= kPatch) describe_cell_tab(tid, win, scale, s_cell);
    __syncthreads();
    if (!area_fast) describe_pix_tab(tid, NT, win, scale, s_cell, s_pix);
    __syncthreads();

    // ---------------------------------------------------------------- descriptor window (A.6)
    const float descriptor_dir = s_dir;
    const float dir = descriptor_dir * (float)(3.1415926535897932384626433832795 / 180);
    // sin/cos(float) defined as the double function rounded to float (see oracle_surf.cpp)
    const float sin_dir = -(float)sin((double)dir);
    const float cos_dir = (float)cos((double)dir);
    const float win_offset = -(float)(win - 1) / 2;
    float start_x = cx + win_offset * cos_dir + win_offset * sin_dir;
    float start_y = cy - win_offset * sin_dir + win_offset * cos_dir;
    const int ncols1 = W - 1, nrows1 = H - 1;
    // upright: WIN[i][j] = img(clamp(y0 - j), clamp(x0 + i)) with x0 = cvRound(cx + win_offset), y0 = cvRound(cy - win_offset)
    const int up_x0 = d_cvRoundF(cx + win_offset), up_y0 = d_cvRoundF(cy - win_offset);

    // rows of the window: thread <-> row, columns walked sequentially so that the double coordinate
    // accumulation (pixel_x += cos_dir) is reproduced step for step, whatever the values; the row never exists as
    // pixels, it is folded straight into the 21 horizontal accumulators of the INTER_AREA row pass.
#pragma unroll 1
    for (int r = 0; r < tid; r++) {
        start_x += sin_dir;
        start_y += cos_dir;
    }
    int* s_rowint = reinterpret_cast<int*>(s_rowbuf);
    const double dcos = (double)cos_dir, dsin = (double)sin_dir;
    for (int base = 0; base < win; base += NT) {
        const int i = base + tid;
        if (i < win) {
            double px = (double)start_x, py = (double)start_y;
            float facc = 0.f;
            int iacc = 0, c = 0, left = iscale;
            for (int j = 0; j < win; j++) {
                const int Sv = upright ? (int)img[(size_t)min(max(up_y0 - j, 0), nrows1) * W + min(max(up_x0 + i, 0), ncols1)]
                                       : sample_pixel(img, W, ncols1, nrows1, px, py);
                px += dcos;
                py -= dsin;
                if (area_fast) {
                    iacc += Sv;
                    if (--left == 0) {
                        s_rowint[i * (kPatch + 1) + c] = iacc;
                        c++;
                        iacc = 0;
                        left = iscale;
                    }
                } else {
                    const PixTab t = s_pix[j];
                    const float Sf = (float)Sv;
                    facc += Sf * t.w_cur;
                    if (t.end) {
                        s_rowbuf[i * (kPatch + 1) + c] = facc;
                        c++;
                        facc = Sf * t.w_next;
                    }
                }
            }
        }
#pragma unroll 1
        for (int r = 0; r < NT; r++) {  // this thread's next row is NT rows further down
            start_x += sin_dir;
            start_y += cos_dir;
        }
    }
    __syncthreads();

    // vertical pass of the decimation: one destination pixel per work item
    for (int it = tid; it < (kPatch + 1) * (kPatch + 1); it += NT) {
        const int dy = it / (kPatch + 1), dxc = it - dy * (kPatch + 1);
        int out;
        if (area_fast) {
            int tot = 0;
            for (int sy = dy * iscale; sy < (dy + 1) * iscale; sy++) tot += s_rowint[sy * (kPatch + 1) + dxc];
            if (iscale == 1)
                out = tot;
            else if (iscale == 2)
                out = (tot + 2) >> 2;
            else
                out = saturate_u8(d_cvRoundF((float)tot * (1.f / (float)(iscale * iscale))));
        } else {
            const CellTab c = s_cell[dy];
            float sumv = 0.f;
            if (c.has_head) sumv += c.a_head * s_rowbuf[(c.sx1 - 1) * (kPatch + 1) + dxc];
            for (int sy = c.sx1; sy < c.sx2; sy++) sumv += c.a_mid * s_rowbuf[sy * (kPatch + 1) + dxc];
            if (c.has_tail) sumv += c.a_tail * s_rowbuf[c.sx2 * (kPatch + 1) + dxc];
            out = saturate_u8(d_cvRoundF(sumv));
        }
        s_patch[it] = (uint8_t)out;
    }
    __syncthreads();

    if (extended)
        describe_patch_to_vec_ext(tid, NT, s_patch, s_rowbuf, s_vec, desc_base + ((size_t)frame * cap + k) * 128);
    else
        describe_patch_to_vec(tid, NT, s_patch, s_rowbuf, s_vec, desc_base + ((size_t)frame * cap + k) * 64);
    }  // work queue
}

// Two keypoints per wavefront (one per 32-lane half), for the stand-alone launch.  The 72 windows of a keypoint are the
// reference's loop i = 0, 5, ... 355; lane L < 24 of a half sums windows 5L, 5L + 120 and 5L + 240, each decided by the
// reference's own test on |A - i| (one v_sad_u32 and one compare per window), i.e. 9 VALU instructions per sample for two
// keypoints where the 36-lane form above spends 6 for one.  Samples past a keypoint's count are (+0, +0): adding them
// changes no bit of a sum.
__device__ __forceinline__ void orientation_pair(const int32_t* __restrict__ sum, int sw, int H, int W, vo_keypoint* kpp, bool valid,
                                                 int lane, float2* s_XY, int* s_A) {
    const int l32 = lane & 31;
    // the sample pattern of this lane's four passes - the same for every keypoint - is requested before the keypoint record
    // (indexed per lane, the constant table comes through vector loads: one round trip less in front of every pass) and the
    // passes are unrolled: 1.70 -> 1.55 ms per 128 frames.  (The taps of pass p + 1 requested before the arithmetic of pass p
    // as well: 1.54, not kept.)
    float tpx[4], tpy[4], tpw[4];
#pragma unroll
    for (int pass = 0; pass < 4; pass++) {
        const int kk = min(pass * 32 + l32, kOriSamples - 1);
        tpx[pass] = c_tab.aptx[kk], tpy[pass] = c_tab.apty[kk], tpw[pass] = c_tab.aptw[kk];
    }
    const float cx = valid ? kpp->x : 0.f, cy = valid ? kpp->y : 0.f;
    const float s = (valid ? kpp->size : 9.f) * 1.2f / 9.0f;
    const int g = 2 * d_cvRoundF(2 * s);
#pragma unroll
    for (int q = 0; q < 4; q++) {  // 128 slots per half
        s_XY[q * 32 + l32] = make_float2(0.f, 0.f);
        s_A[q * 32 + l32] = 0;
    }
    wave_sync_lds();
    int nangle = 0;
    const bool fits = valid && !(H + 1 < g || W + 1 < g);
    {
        const int xm = g / 2;
        const float wxa = -1.f / ((float)xm * g), wxb = 1.f / ((float)(g - xm) * g);
        const float wya = 1.f / ((float)g * xm), wyb = -1.f / ((float)g * (g - xm));
        const float half = (float)(g - 1) / 2;
#pragma unroll
        for (int pass = 0; pass < 4; pass++) {
            const int kk = pass * 32 + l32;
            bool inb = false;
            float X = 0, Y = 0;
            int Ar = 0;
            if (fits && kk < kOriSamples) {
                const int x = d_cvRoundF(cx + tpx[pass] * s - half);
                const int y = d_cvRoundF(cy + tpy[pass] * s - half);
                inb = !(y < 0 || y >= H + 1 - g || x < 0 || x >= W + 1 - g);
                if (inb) {
                    const int32_t* p = sum + (size_t)y * sw + x;
                    const uint32_t t00 = p[0], t0m = p[xm], t0g = p[g];
                    const uint32_t tm0 = p[(size_t)xm * sw], tmg = p[(size_t)xm * sw + g];
                    const uint32_t tg0 = p[(size_t)g * sw], tgm = p[(size_t)g * sw + xm], tgg = p[(size_t)g * sw + g];
                    const int bxa = (int)(t00 + tgm - tg0 - t0m), bxb = (int)(t0m + tgg - tgm - t0g);
                    double d = (double)((float)bxa * wxa);
                    d += (double)((float)bxb * wxb);
                    const float vx = (float)d;
                    const int bya = (int)(t00 + tmg - tm0 - t0g), byb = (int)(tm0 + tgg - tg0 - tmg);
                    d = (double)((float)bya * wya);
                    d += (double)((float)byb * wyb);
                    const float vy = (float)d;
                    const float w = tpw[pass];
                    X = vx * w;
                    Y = vy * w;
                    Ar = d_cvRoundF(d_fastAtan2(Y, X));
                }
            }
            const unsigned long long m = __ballot(inb);
            const unsigned hm = (unsigned)(m >> (lane & 32));  // this half's lanes
            if (inb) {
                const int pos = nangle + __popc(hm & ((1u << l32) - 1u));
                s_XY[pos] = make_float2(X, Y);
                s_A[pos] = Ar;
            }
            nangle += __popc(hm);
        }
    }
    wave_sync_lds();
    // trip count: the larger of the two halves' counts (a scalar)
    const int n_u = max(__builtin_amdgcn_readlane(nangle, 0), __builtin_amdgcn_readlane(nangle, 32));
    const int i0 = (l32 < 24 ? l32 : 0) * 5, i1 = i0 + 120, i2 = i0 + 240;
    typedef float f32x2_t __attribute__((ext_vector_type(2)));
    f32x2_t acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, acc2 = {0.f, 0.f};
    auto window_step = [&](int j) {
        const int A = s_A[j];
        const float2 xyv = s_XY[j];
        const f32x2_t xy = {xyv.x, xyv.y};
        unsigned d0, d1, d2;  // |A - i| - 30: in the window iff < 0 (wraps) or > 300
        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d0) : "v"(A), "v"(i0), "v"(-30));
        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d1) : "v"(A), "v"(i1), "v"(-30));
        asm("v_sad_u32 %0, %1, %2, %3" : "=v"(d2) : "v"(A), "v"(i2), "v"(-30));
        if (d0 > 300u) {
            acc0 += xy;
            asm volatile("" : "+v"(acc0));
        }
        if (d1 > 300u) {
            acc1 += xy;
            asm volatile("" : "+v"(acc1));
        }
        if (d2 > 300u) {
            acc2 += xy;
            asm volatile("" : "+v"(acc2));
        }
    };
    int j = 0;
    for (; j + 4 <= n_u; j += 4) {
        window_step(j);
        window_step(j + 1);
        window_step(j + 2);
        window_step(j + 3);
    }
    for (; j < n_u; j++) window_step(j);
    float best_mod = 0.f, best_x = 0.f, best_y = 0.f;
    int best_i = 1 << 30;
    if (l32 < 24) {  // ascending i: strictly greater only
        const float m0 = acc0.x * acc0.x + acc0.y * acc0.y;
        if (m0 > best_mod) best_mod = m0, best_x = acc0.x, best_y = acc0.y, best_i = i0;
        const float m1 = acc1.x * acc1.x + acc1.y * acc1.y;
        if (m1 > best_mod) best_mod = m1, best_x = acc1.x, best_y = acc1.y, best_i = i1;
        const float m2 = acc2.x * acc2.x + acc2.y * acc2.y;
        if (m2 > best_mod) best_mod = m2, best_x = acc2.x, best_y = acc2.y, best_i = i2;
    }
#pragma unroll
    for (int d = 16; d >= 1; d >>= 1) {  // within the half: larger mod wins, ties go to the smaller i
        const float om = __shfl_xor(best_mod, d, 64), ox = __shfl_xor(best_x, d, 64), oy = __shfl_xor(best_y, d, 64);
        const int oi = __shfl_xor(best_i, d, 64);
        if (om > best_mod || (om == best_mod && oi < best_i)) best_mod = om, best_x = ox, best_y = oy, best_i = oi;
    }
    if (!(best_mod > 0.f)) best_x = 0.f, best_y = 0.f;
    const float dd = d_fastAtan2(-best_y, best_x);
    if (valid && l32 == 0) kpp->angle = dd;
}

// a5 as its own launch: two keypoints per wavefront (orientation_pair), eight per workgroup.  Keeping it out of the descriptor
// kernel takes its registers and LDS off that kernel's budget and leaves no wavefront idle while one computes the angle.
// upright: every keypoint's angle is 360 - 90 degrees (SURFInvoker), no orientation search
__global__ __launch_bounds__(256) void k_upright_angle(vo_keypoint* __restrict__ kp_base, int cap, const FrameCounters* __restrict__ cnt) {
    const int f = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < min(cnt[f].n_kp, cap)) kp_base[(size_t)f * cap + k].angle = 360.f - 90.f;
}

__global__ __launch_bounds__(256) void k_orientation(const int32_t* __restrict__ sum_base, size_t sum_stride, int H, int W,
                                                     vo_keypoint* __restrict__ kp_base, int cap,
                                                     const FrameCounters* __restrict__ cnt) {
    __shared__ float2 s_XY[8][128];
    __shared__ int s_A[8][128];
    const int frame = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, hw = wave * 2 + (lane >> 5);
    const int n = min(cnt[frame].n_raw, cap);
    if ((blockIdx.x * 4 + wave) * 2 >= n) return;  // neither half has a keypoint
    const int k = blockIdx.x * 8 + hw;
    const bool valid = k < n;
    vo_keypoint* kpp = kp_base + (size_t)frame * cap + (valid ? k : 0);
    orientation_pair(sum_base + (size_t)frame * sum_stride, W + 1, H, W, kpp, valid, lane, s_XY[hw], s_A[hw]);
}

// Bilinear footprint image for the descriptor sampler: one aligned 32-bit word per pixel holds the 2x2 taps
// (p00 | p01 << 8 | p10 << 16 | p11 << 24), so a sample costs ONE gather instead of two unaligned 16-bit ones
// (tools/microbench/gather_bench.hip: 2.2-2.5x the gather rate of the u8 image on gfx950).  The last row and
// column repeat the edge pixel; the sampler only reads footprints with ix < W-1, iy < H-1.
// Layout of the footprint image: 8-pixel-wide COLUMN STRIPS, strip after strip, rows consecutive inside a strip - the word of
// pixel (y, x) sits at quad_index(H, y, x).  A 128-byte cache line then holds an 8 x 4 block of pixels instead of 32 x 1, and
// a gather instruction of the descriptor sampler (8 window rows x 8 columns, rotated: an ~11 x 11 pixel patch) touches ~8
// lines instead of ~15.  The sampler is bound by exactly that - knock-outs over the four window classes, ms per 128 frames:
// 19.9 as it was, 15.4 without the loads, 19.2 without the bilinear arithmetic, 17.3 with every load made to hit L1 - and
// the strips bring it to 18.4 (widths of 4 and 16 pixels: 18.5).
__host__ __device__ __forceinline__ size_t quad_stride(int H, int W) { return (size_t)((W + 7) >> 3) * 8 * H; }
__device__ __forceinline__ size_t quad_index(int H, int y, int x) { return (size_t)(x >> 3) * (8 * H) + (size_t)y * 8 + (x & 7); }

__global__ __launch_bounds__(256) void k_quads(const uint8_t* __restrict__ img_base, size_t img_stride, int H, int W,
                                               uint32_t* __restrict__ quad_base) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, frame = blockIdx.z;
    if (x >= W) return;
    const uint8_t* img = img_base + (size_t)frame * img_stride;
    const int x1 = min(x + 1, W - 1), y1 = min(y + 1, H - 1);
    const uint32_t p00 = img[(size_t)y * W + x], p01 = img[(size_t)y * W + x1];
    const uint32_t p10 = img[(size_t)y1 * W + x], p11 = img[(size_t)y1 * W + x1];
    quad_base[(size_t)frame * quad_stride(H, W) + quad_index(H, y, x)] = p00 | (p01 << 8) | (p10 << 16) | (p11 << 24);
}

// the same for frames whose rows are 4-byte aligned (W and the frame pitch multiples of four): four pixels per lane from two
// aligned words + the next byte of each row, one 16-byte store.  Eight consecutive lanes fill one 128-byte line of a strip
// (two lanes per row, four rows), a wavefront eight strips of those four rows.
__global__ __launch_bounds__(256) void k_quads4(const uint8_t* __restrict__ img_base, size_t img_stride, int H, int W,
                                                uint32_t* __restrict__ quad_base) {
    const int t = threadIdx.x, frame = blockIdx.z;
    const int strip = blockIdx.x * 32 + (t >> 3), y = blockIdx.y * 4 + ((t >> 1) & 3), x = strip * 8 + (t & 1) * 4;
    if (x >= W || y >= H) return;
    const uint8_t* img = img_base + (size_t)frame * img_stride;
    const uint8_t* r0 = img + (size_t)y * W + x;
    const uint8_t* r1 = img + (size_t)min(y + 1, H - 1) * W + x;
    const uint32_t a = *reinterpret_cast<const uint32_t*>(r0), b = *reinterpret_cast<const uint32_t*>(r1);
    const bool last = x + 4 >= W;  // the frame's last column repeats itself
    const uint32_t an = last ? a >> 24 : r0[4], bn = last ? b >> 24 : r1[4];
    const uint64_t a64 = (uint64_t)a | ((uint64_t)an << 32), b64 = (uint64_t)b | ((uint64_t)bn << 32);
    uint4 q;
    q.x = (uint32_t)(a64 & 0xffffu) | ((uint32_t)(b64 & 0xffffu) << 16);
    q.y = (uint32_t)((a64 >> 8) & 0xffffu) | ((uint32_t)((b64 >> 8) & 0xffffu) << 16);
    q.z = (uint32_t)((a64 >> 16) & 0xffffu) | ((uint32_t)((b64 >> 16) & 0xffffu) << 16);
    q.w = (uint32_t)((a64 >> 24) & 0xffffu) | ((uint32_t)((b64 >> 24) & 0xffffu) << 16);
    *reinterpret_cast<uint4*>(quad_base + (size_t)frame * quad_stride(H, W) + quad_index(H, y, x)) = q;
}

// ------------------------------------------------------------------------------------------------
// a6, main path: descriptor window sampled in compact 2-D blocks.
//
// The reference walks every window row with pixel_x += cos_dir in double (A.6).  start_x and cos_dir are
// floats, so whenever |cos_dir|, |sin_dir| and the row start are 0 or >= 2^-17 (and < 7000) every partial
// sum start + j*step is a multiple of 2^-40 below 2^13: it fits a double exactly, the sequential
// accumulation never rounds, and pixel(i, j) = fma(j, step, start_i) is the SAME double.  That frees the lane
// mapping: a wavefront samples RB rows x 64/RB consecutive columns per load instruction, so the 64 bilinear
// footprints of one gather touch ~10 cache lines instead of ~45 (row-per-lane walk).  Samples go to a u8
// strip in LDS; the INTER_AREA row pass is one (row, destination cell) per lane, the vertical pass streams
// strip by strip through 21 lanes, so the window never exists outside LDS.  A keypoint that fails the
// exactness test (e.g. an orientation of exactly 90 degrees: cos = -4.4e-8) goes to list 4 and is described by
// the sequential walk of k_describe.
// ------------------------------------------------------------------------------------------------
// sin/cos(float) are defined as the double functions rounded to float (see oracle_surf.cpp).  Out of line: the
// double-precision argument reduction would otherwise set the register budget of the whole kernel.
__device__ __noinline__ float2 direction_sincos(float dir) {
    double sd, cd;
    sincos((double)dir, &sd, &cd);
    return make_float2((float)sd, (float)cd);
}

__device__ __forceinline__ bool exact_step(float v) { return v == 0.f || fabsf(v) >= 0x1p-17f; }
__device__ __forceinline__ bool exact_start(float v) { return (v == 0.f || fabsf(v) >= 0x1p-17f) && fabsf(v) < 7000.f; }

typedef float f32x2 __attribute__((ext_vector_type(2)));

// bilinear sample from one footprint word (k_quads), rounded to u8;
// the reference's expression order: p00*(1-a)*(1-b) + p01*a*(1-b) + p10*(1-a)*b + p11*a*b
__device__ __forceinline__ int bilinear_u8(uint32_t q, float a, float b) {
    const f32x2 wx = {1.f - a, a};
    const float omb = 1.f - b;
    f32x2 r0 = {(float)(q & 255u), (float)((q >> 8) & 255u)};
    f32x2 r1 = {(float)((q >> 16) & 255u), (float)(q >> 24)};
    r0 = r0 * wx;
    r1 = r1 * wx;
    r0 = r0 * omb;
    r1 = r1 * b;
    float v = r0.x + r0.y;
    v = v + r1.x;
    v = v + r1.y;
    // v_cvt_pk_u8_f32: round to nearest even + saturate to u8 in one instruction (v is in [0, 255], so the
    // saturation never acts and this is cvRound followed by the uchar cast)
    return (int)__builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u);
}

// U samples of one lane at columns j0 + u * CB, all inside the window, walked in double precision.  (px0, py0) is the exact
// position of column j0, (dxs, dys) the exact step of CB columns.  Coordinates are linear in the column, so the batch lies
// in the frame interior iff its first and last sample do; then the U footprint loads go out back to back.  (The rows of
// windows inside the frame whose coordinates are multiples of 2^-32 - nearly all - take sample_batch_fix instead.)
template <int U, int CB>
__device__ __forceinline__ void sample_batch(const uint8_t* __restrict__ img, const uint32_t* __restrict__ quad, int W, int qs8,
                                             int ncols1, int nrows1, double px0, double py0, double dxs, double dys,
                                             lds_u8* wdst) {
    const double pxl = px0 + (double)(U - 1) * dxs, pyl = py0 + (double)(U - 1) * dys;
    const int ix0 = (int)floor(px0), iy0 = (int)floor(py0), ixl = (int)floor(pxl), iyl = (int)floor(pyl);
    const bool ok = (unsigned)ix0 < (unsigned)ncols1 && (unsigned)iy0 < (unsigned)nrows1 && (unsigned)ixl < (unsigned)ncols1 &&
                    (unsigned)iyl < (unsigned)nrows1;
    if (__ballot(!ok) == 0ull) {
        float a[U], b[U];
        uint32_t q[U];
        double px = px0, py = py0;
#pragma unroll
        for (int u = 0; u < U; u++) {
            // interior samples have px, py >= 0: truncation is floor, and fract(p) = p - floor(p) exactly
            a[u] = (float)__builtin_amdgcn_fract(px);
            b[u] = (float)__builtin_amdgcn_fract(py);
            asm volatile("" : "+v"(a[u]), "+v"(b[u]));  // keep the fractions as floats (not as doubles) across the loads
            // 32-bit byte offset from the frame's base (a frame's footprint image is < 4 GB): scalar base + vector offset.
            // The footprint image is stored in 8-pixel-wide column strips (quad_index): x >> 3 picks the strip, (y, x & 7) the word
            const uint32_t xi = (uint32_t)(int)px, yi = (uint32_t)(int)py;
            // byte offset 4 * ((x >> 3) * 8H + 8y + (x & 7)) = (x >> 3) * (32H - 32) + 32y + 4x: four instructions
            uint32_t boff = (uint32_t)__mul24((int)(xi >> 3), qs8);
            asm("v_lshl_add_u32 %0, %1, 5, %0" : "+v"(boff) : "v"(yi));  // (the compiler's choice is two shifts and a three-way add)
            asm("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(boff) : "v"(xi));
            q[u] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(quad) + boff);
            px += dxs;
            py += dys;
        }
#pragma unroll
        for (int u = 0; u < U; u++) wdst[u * CB] = (uint8_t)bilinear_u8(q[u], a[u], b[u]);
    } else {  // frame border: the generic sampler (clamped nearest pixel outside the interior)
        double px = px0, py = py0;
#pragma unroll 1
        for (int u = 0; u < U; u++) {
            wdst[u * CB] = (uint8_t)sample_pixel(img, W, ncols1, nrows1, px, py);
            px += dxs;
            py += dys;
        }
    }
}

// ---- 32.32 fixed-point walk of the interior samples ----------------------------------------------------------------------
// When cos, sin and a row's start are 0 or >= 2^-9 in magnitude (floats: multiples of 2^-32) every coordinate start + j * step of
// the row is a multiple of 2^-32 below 2^13, i.e. an exact 32.32 fixed-point number: one 64-bit add steps it, the high word is
// floor(p), and v_cvt_f32_u32 of the low word is 2^32 times the reference's (float)(p - floor(p)) - the same rounding, scaled
// by a power of two.  The bilinear weights carry that factor (2^32 - A is 2^32 * fl(1 - a); products and sums scale exactly as long
// as nothing under- or overflows, see bilinear_u8_fix).  No double-precision arithmetic in the loop.
__device__ __forceinline__ bool exact32(float v) { return v == 0.f || fabsf(v) >= 0x1p-9f; }
__device__ __forceinline__ uint64_t fix32_from_float(float v) {  // v >= 0, a multiple of 2^-32 below 2^32 (v == 0 or v >= 2^-9)
    const uint32_t hi = (uint32_t)v;  // v >= 0: truncation is floor
    const uint32_t lo = (uint32_t)(__builtin_amdgcn_fractf(v) * 4294967296.f);  // v - floor(v) and the scaling are exact
    return ((uint64_t)hi << 32) | lo;
}
// one instruction; the step is wave-uniform and stays in an SGPR pair (a VGPR pair would make the add full rate, but the 96-register
// classes have no registers to spare: measured 17.80 -> 17.60 ms over the four classes with the step in SGPRs)
// (the SGPR pairs it reads are written by v_readfirstlane at the top of a keypoint, hundreds of instructions before the first add:
// the two wait states a VALU read of a VALU-written SGPR needs are not the compiler's to insert around inline assembly)
__device__ __forceinline__ uint64_t add_u64(uint64_t a, uint64_t b_uniform) {
    uint64_t r;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(r) : "v"(a), "s"(b_uniform));
    return r;
}
// byte K of q, taken as the float whose bits it is (the denormal byte * 2^-149), times w: ONE instruction where the conversion
// and the product would be two.  The f32 denormal mode of these kernels is IEEE, so the product is the correctly rounded
// byte * w * 2^-149 as long as it is a normal number - the caller's scaling sees to that.
template <int K>
__device__ __forceinline__ float mul_byte_denorm(uint32_t q, float w) {
    float r;
    static_assert(K >= 0 && K < 4, "byte index");
    if (K == 0) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD" : "=v"(r) : "v"(q), "v"(w));
    if (K == 1) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "=v"(r) : "v"(q), "v"(w));
    if (K == 2) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD" : "=v"(r) : "v"(q), "v"(w));
    if (K == 3) asm("v_mul_f32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "=v"(r) : "v"(q), "v"(w));
    return r;
}
// A = 2^32 a, B = 2^32 b.  The x weights are carried at 2^117 (= 2^32 * 2^85): a pixel byte read as a denormal is p * 2^-149, so
// the first product is 2^-32 * fl(p * (1 - a)) (between 2^-56 and 2^-24: normal), the second, by 2^32 (1 - b) or 2^32 b, is the
// reference's own value, and the sum needs no rescaling.
__device__ __forceinline__ int bilinear_u8_fix(uint32_t q, float A, float B) {
    const float w1 = A * 0x1p85f, w0 = 0x1p117f - w1;  // exact scalings of a and of fl(1 - a)
    const float omb = 4294967296.f - B;
    f32x2 r0 = {mul_byte_denorm<0>(q, w0), mul_byte_denorm<1>(q, w1)};
    f32x2 r1 = {mul_byte_denorm<2>(q, w0), mul_byte_denorm<3>(q, w1)};
    r0 = r0 * omb;
    r1 = r1 * B;
    float v = r0.x + r0.y;
    v = v + r1.x;
    v = v + r1.y;
    return (int)__builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u);
}
// (the compiler drops a readfirstlane of a value it knows to be uniform and then keeps that value in VGPRs, also as the "s"
// operand of inline assembly; the empty asm hides where the value comes from.  The readfirstlane itself stays a builtin: an
// SGPR written by a VALU instruction needs two wait states before a VALU instruction reads it on this part, and the
// compiler's hazard pass does not look inside inline assembly)
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
    asm volatile("" : "+v"(lo), "+v"(hi));
    lo = __builtin_amdgcn_readfirstlane(lo);
    hi = __builtin_amdgcn_readfirstlane(hi);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double uniform_f64(double v) { return __builtin_bit_cast(double, uniform_u64(__builtin_bit_cast(uint64_t, v))); }
template <int U, int CB>
__device__ __forceinline__ void sample_batch_fix(const uint32_t* __restrict__ quad, int qs8, uint64_t& X, uint64_t& Y, uint64_t DX,
                                                 uint64_t DY, lds_u8* wdst) {
    float A[U], B[U];
    uint32_t q[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
        A[u] = (float)(uint32_t)X;
        B[u] = (float)(uint32_t)Y;
        const uint32_t xi = (uint32_t)(X >> 32), yi = (uint32_t)(Y >> 32);
        uint32_t boff = (uint32_t)__mul24((int)(xi >> 3), qs8);
        asm("v_lshl_add_u32 %0, %1, 5, %0" : "+v"(boff) : "v"(yi));
        asm("v_lshl_add_u32 %0, %1, 2, %0" : "+v"(boff) : "v"(xi));
        q[u] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const unsigned char*>(quad) + boff);
        X = add_u64(X, DX);
        Y = add_u64(Y, DY);
    }
#pragma unroll
    for (int u = 0; u < U; u++) wdst[u * CB] = (uint8_t)bilinear_u8_fix(q[u], A[u], B[u]);
}

// The decimation tables depend on the window size only: built once per context for every size the tiled kernels see
// (k_build_win_tables, the same device functions as the sequential kernel uses), then loaded per keypoint instead of being
// recomputed (22 double divisions + a 4-cell search per source pixel, ~200-1400 wavefront instructions per keypoint).
constexpr int kWinTabPitch = kBigWin;  // PixTab entries per window size
__global__ __launch_bounds__(256) void k_build_win_tables(CellTab* __restrict__ g_cell, PixTab* __restrict__ g_pix) {
    __shared__ CellTab s_cell[kPatch + 1];
    __shared__ PixTab s_pix[kBigWin];
    const int win = blockIdx.x + 1, tid = threadIdx.x;
    const double scale = (double)win / (kPatch + 1);
    if (tid <= kPatch) describe_cell_tab(tid, win, scale, s_cell);
    __syncthreads();
    describe_pix_tab(tid, 256, win, scale, s_cell, s_pix);
    __syncthreads();
    if (tid <= kPatch) {
        // smallest and largest number of full-weight pixels of a cell of this window size: the bounds of the row pass's
        // unmasked and masked groups (k_describe_tiled), the same in every lane
        int nmin = 1 << 15, nmax = 0;
        for (int c = 0; c <= kPatch; c++) {
            const int n = s_cell[c].sx2 - s_cell[c].sx1;
            nmin = min(nmin, n);
            nmax = max(nmax, n);
        }
        CellTab ct = s_cell[tid];
        ct.pad = nmin | (nmax << 16);
        g_cell[(size_t)win * (kPatch + 1) + tid] = ct;
    }
    for (int j = tid; j < win; j += 256) g_pix[(size_t)win * kWinTabPitch + j] = s_pix[j];
}

// The descriptor sampler and the row pass multiply pixel bytes as DENORMAL floats (mul_byte_denorm): correct only while the
// kernels run with IEEE f32 denormals (the compiler's default for HIP; -fgpu-flush-denormals-to-zero or a fast-math build
// would silently turn every product into zero).  Checked once per context, on the device, before anything depends on it.
__global__ __launch_bounds__(64) void k_denorm_selfcheck(int* __restrict__ ok) {
    const uint32_t t = threadIdx.x;
    const uint32_t q = t | ((t + 64u) << 8) | ((t + 128u) << 16) | ((t + 192u) << 24);
    const float w = 0.75f * 0x1p127f;
    const bool good = mul_byte_denorm<0>(q, w) * 0x1p22f == (float)t * 0.75f && mul_byte_denorm<1>(q, w) * 0x1p22f == (float)(t + 64u) * 0.75f &&
                      mul_byte_denorm<2>(q, w) * 0x1p22f == (float)(t + 128u) * 0.75f &&
                      mul_byte_denorm<3>(q, w) * 0x1p22f == (float)(t + 192u) * 0.75f;
    const unsigned long long m = __ballot(good);
    if (t == 0) *ok = m == ~0ull ? 1 : 0;
}

int vo_build_window_tables(vo_ctx* ctx) {
    // the row pass of k_describe_tiled handles ONE group of four stream bytes in which the cells of a window differ: check that
    // no window size needs more (the arithmetic of describe_cell_tab, on the host)
    for (int win = 1; win <= kBigWin; win++) {
        const double scale = (double)win / (kPatch + 1);
        int nmin = 1 << 30, nmax = 0;
        for (int c = 0; c <= kPatch; c++) {
            const double fsx1 = c * scale, fsx2 = fsx1 + scale;
            int sx1 = (int)std::ceil(fsx1), sx2 = (int)std::floor(fsx2);
            sx2 = std::min(sx2, win - 1);
            sx1 = std::min(sx1, sx2);
            nmin = std::min(nmin, sx2 - sx1);
            nmax = std::max(nmax, sx2 - sx1);
        }
        const int jm = std::max(4, 4 * ((nmin + 1) >> 2));
        if (jm + 3 < nmax) {
            fprintf(stderr, "vo_build_window_tables: window %d has cells of %d..%d pixels, more than one ragged group\n", win, nmin, nmax);
            abort();
        }
    }
    VO_HIP(hipMalloc(&ctx->d_win_cell, (size_t)(kBigWin + 1) * (kPatch + 1) * sizeof(CellTab)));
    VO_HIP(hipMalloc(&ctx->d_win_pix, (size_t)(kBigWin + 1) * kWinTabPitch * sizeof(PixTab)));
    hipLaunchKernelGGL(k_build_win_tables, dim3(kBigWin), dim3(256), 0, ctx->stream, (CellTab*)ctx->d_win_cell, (PixTab*)ctx->d_win_pix);
    VO_HIP(hipGetLastError());
    // (the first word of the pixel table doubles as the self-check's flag: window size 0 has no entries)
    hipLaunchKernelGGL(k_denorm_selfcheck, dim3(1), dim3(64), 0, ctx->stream, (int*)ctx->d_win_pix);
    VO_HIP(hipGetLastError());
    int denorm_ok = 0;
    VO_HIP(hipMemcpyAsync(&denorm_ok, ctx->d_win_pix, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    VO_HIP(hipStreamSynchronize(ctx->stream));
    if (!denorm_ok) {
        vo_set_error_text("f32 denormals are flushed in this build of the library: the descriptor kernels need IEEE denormals");
        return VO_ERR_UNSUPPORTED;
    }
    return VO_OK;
}

// WHI: largest window of the class; NW wavefronts per keypoint; a gather covers RB rows x 64/RB columns; UMAX gathers in
// flight per lane; a strip is G groups of RB rows per wavefront (sampled back to back, then one row pass and one fold);
// CH work-queue tickets per atomic.
template <int WHI, int NW, int RB, int UMAX, int G, int CH>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(WHI > 320 ? 4 : 5, 8))) void k_describe_tiled(const uint8_t* __restrict__ img_base, size_t img_stride,
                                                            const uint32_t* __restrict__ quad_base,
                                                            const int32_t* __restrict__ sum_base, size_t sum_stride, int H,
                                                            int W, vo_keypoint* __restrict__ kp_base,
                                                            float* __restrict__ desc_base, int cap,
                                                            const int* __restrict__ list, int* __restrict__ cls_cnt, int cls,
                                                            int* __restrict__ seq_list, int force_seq,
                                                            const CellTab* __restrict__ g_cell, const PixTab* __restrict__ g_pix,
                                                            const float2* __restrict__ dir_sc) {
    constexpr int NT = NW * 64;
    constexpr int CB = 64 / RB;                    // columns of one gather
    constexpr int WR = RB * G;                     // rows of one wavefront in a strip
    constexpr int RS = NW * WR;                    // rows of one strip
    constexpr int NBUF = NW > 1 ? 2 : 1;           // row-pass results are double-buffered only across wavefronts
    constexpr int kPitch = ((WHI + 3) & ~3) + 4;   // bytes per strip row: an odd number of dwords
    static_assert(((kPitch / 4) & 1) == 1, "strip pitch must be an odd number of dwords");
    __shared__ CellTab s_cell[kPatch + 1];
    // strip-phase arrays; the gradient planes of the last phase (s_grad, 3200 B) reuse the same bytes
    constexpr int kOffSx = 0, kOffSy = kOffSx + WHI * 4;
    constexpr int kOffWin = (kOffSy + WHI * 4 + 15) & ~15, kOffRowv = (kOffWin + RS * kPitch + 15) & ~15;
    constexpr int kStripBytes = kOffRowv + NBUF * RS * (kPatch + 1) * 4, kGradBytes = 2 * kPatch * kPatch * 4;
    // (+ 256 bytes: the vertical fold reads up to three rows past the last row of the strip buffer and discards them)
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[(kStripBytes > kGradBytes ? kStripBytes : kGradBytes) + 256];
    float* s_sx = reinterpret_cast<float*>(s_raw + kOffSx);   // start of every window row (sequential float accumulation)
    float* s_sy = reinterpret_cast<float*>(s_raw + kOffSy);
    uint8_t* s_win = s_raw + kOffWin;
    float* s_rowv = reinterpret_cast<float*>(s_raw + kOffRowv);  // row-pass results of a strip (float, or int for integer scales)
    float* s_grad = reinterpret_cast<float*>(s_raw);
    __shared__ uint8_t s_patch[(kPatch + 1) * (kPatch + 1) + 7];
    __shared__ float s_vec[64];
    __shared__ int s_q[8];  // the keypoint being described: ticket, frame * cap + k, then x, y, size, angle as bits

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int r = lane / CB, c = lane - r * CB;
    const int n_items = cls_cnt[cls];
    // Work queue with a one-keypoint look-ahead: thread 0 draws the NEXT ticket, list entry and keypoint record
    // while the workgroup describes the current one, so the three dependent global round trips (atomic, list,
    // record) are off the critical path of a keypoint.
    // Tickets are drawn CH at a time (atomics on one counter complete at ~12 ns each on this part, which would otherwise
    // bound the classes of small windows).  The list is cut into 8 contiguous segments with a counter each; a workgroup
    // starts in segment blockIdx.x % 8 - the label shared by the workgroups of one XCD - and moves on to the other
    // segments when its own is empty.
    int q_next = 0, q_end = 0, q_seg = blockIdx.x & 7, q_tried = 0;  // thread 0 only
    // a list shorter than the grid's appetite (a single frame pair) is handed out one keypoint at a time: every keypoint gets
    // a workgroup of its own and the launch lasts one keypoint, not CH of them in a row
    const int chunk = n_items >= (int)gridDim.x * CH ? CH : 1;
    // More workgroups than keypoints: the surplus leaves at once, the others drain the queue.  (The eight segment counters of
    // a class share a cache line, and atomics on one line complete at ~11 ns each: the eight failing draws of every one of
    // 5120 workgroups made an EMPTY launch last 0.47 ms - most of a single frame pair's detection time.  A look at the
    // counter before each draw removes them for full launches too, but measured 1.5 % slower there.)
    if ((int)blockIdx.x >= n_items) return;
    auto draw = [&]() -> int {
        if (q_next < q_end) return q_next++;
        while (q_tried < 8) {
            const int lo = (int)((long long)n_items * q_seg >> 3), hi = (int)((long long)n_items * (q_seg + 1) >> 3);
            const int t = lo + atomicAdd(&cls_cnt[ticket_index(cls, q_seg)], chunk);
            if (t < hi) {
                q_next = t + 1;
                q_end = min(t + chunk, hi);
                return t;
            }
            q_seg = (q_seg + 1) & 7;
            q_tried++;
        }
        return n_items;
    };
    if (tid == 0) {
        const int it = draw();
        s_q[0] = it;
        if (it < n_items) {
            const int cd = list[it];
            const vo_keypoint* kq = kp_base + cd;
            s_q[1] = cd;
            s_q[2] = __float_as_int(kq->x);
            s_q[3] = __float_as_int(kq->y);
            s_q[4] = __float_as_int(kq->size);
            s_q[5] = __float_as_int(kq->angle);
            const float2 d = dir_sc[cd];
            s_q[6] = __float_as_int(d.x);
            s_q[7] = __float_as_int(d.y);
        }
    }
    for (;;) {
        __syncthreads();  // s_q is written; the previous keypoint's LDS is no longer read
        const int item = s_q[0];
        if (item >= n_items) break;
        const int code = s_q[1];
        const float cx = __int_as_float(s_q[2]), cy = __int_as_float(s_q[3]), size = __int_as_float(s_q[4]);
        const float sin_dir = -__int_as_float(s_q[6]), cos_dir = __int_as_float(s_q[7]);  // k_classify_count: sin, cos of the angle
        const int frame = code / cap, k = code - frame * cap;
        int n_item = 0, n_code = 0;
        if (tid == 0) n_item = draw();  // look-ahead stage 1: ticket
        const float s = size * 1.2f / 9.0f;
        const int win = (int)((kPatch + 1) * s);
        const uint8_t* __restrict__ img = img_base + (size_t)frame * img_stride;
        const uint32_t* __restrict__ quad = quad_base + (size_t)frame * quad_stride(H, W);
        // scale = win / 21 in double, iscale = cvRound(scale), area_fast = |scale - iscale| < DBL_EPSILON (INTER_AREA's integer-scale
        // branch): win / 21 is an integer iff 21 divides win and never lies halfway between two - in integers, without the division
        const int iscale = (2 * win + (kPatch + 1)) / (2 * (kPatch + 1));
        const bool area_fast = win % (kPatch + 1) == 0;
        // the window size's decimation tables (k_build_win_tables)
        if (!area_fast && tid <= kPatch) s_cell[tid] = g_cell[(size_t)win * (kPatch + 1) + tid];
        if (tid == 0) n_code = *(volatile const int*)&list[min(n_item, n_items - 1)];  // look-ahead stage 2: list entry
        const float win_offset = -(float)(win - 1) / 2;
        const int ncols1 = W - 1, nrows1 = H - 1, qs8 = 32 * H - 32;  // see sample_batch
        // wave-uniform values that live across the sampling loop are pinned to SGPRs (uniform_*): left to itself the compiler
        // keeps them in VGPRs, and the 96-register classes spill for every pair of them
        const double dcos = uniform_f64((double)cos_dir), ndsin = uniform_f64(-(double)sin_dir);
        const double dxs = (double)CB * dcos, dys = (double)CB * ndsin;  // exact: CB is a power of two
        bool bad = force_seq != 0 || !(exact_step(cos_dir) && exact_step(sin_dir));
        const bool fix_ok = exact32(cos_dir) && exact32(sin_dir);
        // the step of CB columns in 32.32 (two's complement: a negative step wraps, which is what the 64-bit add needs)
        const long long fcos = (long long)(dcos * 4294967296.0), fnsin = (long long)(ndsin * 4294967296.0);  // step of one column
        const uint64_t fdx = uniform_u64((uint64_t)(fcos * CB)), fdy = uniform_u64((uint64_t)(fnsin * CB));
        const uint64_t fendx = uniform_u64((uint64_t)(fcos * (win - 1))), fendy = uniform_u64((uint64_t)(fnsin * (win - 1)));  // column 0 -> last column
        const uint64_t fcx = (uint64_t)(fcos * c), fcy = (uint64_t)(fnsin * c);                                           // column 0 -> this lane's
        if (tid >= NT - 2) {
            // row starts are a sequential FLOAT accumulation (start_x += sin_dir): two lanes walk it once per keypoint
            const bool isy = tid == NT - 1;
            float v = isy ? cy - win_offset * sin_dir + win_offset * cos_dir : cx + win_offset * cos_dir + win_offset * sin_dir;
            const float inc = isy ? cos_dir : sin_dir;
            float* tab = isy ? s_sy : s_sx;
#pragma unroll 4
            for (int q = 0; q < win; q++) {
                tab[q] = v;
                v += inc;
            }
        }
        __syncthreads();

        // Is the WHOLE window in the frame interior, with every coordinate exact in 32.32?  Row starts are float accumulations of
        // a constant, hence monotone in the row index: every row's start lies between the first and the last row's, and so does
        // its end (start + (win - 1) * step, exact in double).  The four corners decide for all rows, and the per-group tests
        // below (interior, exactness of the row start: ~45 instructions per group of RB rows, a third of the sampling itself for
        // a 45-pixel window) run only for windows that touch the frame border.
        bool fix_all;
        {
            const float sx0 = s_sx[0], sxn = s_sx[win - 1], sy0 = s_sy[0], syn = s_sy[win - 1];
            const float xlo = fminf(sx0, sxn), xhi = fmaxf(sx0, sxn), ylo = fminf(sy0, syn), yhi = fmaxf(sy0, syn);
            const double ex = (double)(win - 1) * dcos, ey = (double)(win - 1) * ndsin;  // exact
            fix_all = fix_ok && xlo >= 0x1p-9f && ylo >= 0x1p-9f && xhi < 7000.f && yhi < 7000.f && (double)xhi < (double)ncols1 &&
                      (double)yhi < (double)nrows1 && (double)xlo + ex >= 0.0 && (double)xhi + ex < (double)ncols1 &&
                      (double)ylo + ey >= 0.0 && (double)yhi + ey < (double)nrows1;
        }

        lds_u8* wbase = (lds_u8*)s_win + wave * (WR * kPitch);
        int* s_rowi = reinterpret_cast<int*>(s_rowv);
        const int myrow = lane / (kPatch + 1), mycell = lane - myrow * (kPatch + 1);  // row pass: lane = (row mod 3, cell)
        // (the row pass re-reads its cell's table entry from LDS in every strip and derives its weights there: nothing of it is
        // alive across the sampling loop, which has no registers to spare)
        constexpr bool kRpSmall = WHI <= kSmallWin;
        // vertical pass state of lanes 0..20 of the first wavefront (destination column = lane; integer scales)
        int viacc = 0, dy = 0, left = iscale;
        // Vertical pass of the general case: lane (g, column) of the first wavefront, g = 0..2, folds destination rows g, g + 3,
        // g + 6, ... - three destination rows are in flight at once, each added up over ITS source rows in row order, so a
        // strip of RS rows takes about RS / 3 dependent steps instead of RS.  A destination row that straddles two strips keeps
        // its partial sum in the lane (f_carry).  Row weights come from the cell table (the y table is the x table).
        // Only the destination row and its partial sum live across the strips (as vacc / dy did): the row's table entry is
        // re-read from LDS in every strip, the sampling loop has no registers to spare.
        int f_d = tid / (kPatch + 1);  // destination row this lane folds next (> kPatch: done)
        float f_carry = 0.f;           // its partial sum when it began in an earlier strip

        for (int i0 = 0, sidx = 0; i0 < win; i0 += RS, sidx ^= (NBUF - 1)) {
            const int wrow0 = i0 + wave * WR;  // first row of this wavefront's part of the strip
            if (wrow0 < win) {
              for (int g = 0; g < G && wrow0 + g * RB < win; g++) {
                const int grow0 = wrow0 + g * RB;
                // idle rows of the window's last group repeat the group's first row (always inside the window), so
                // that every lane computes a valid sample and no load needs a guard
                const int irow = grow0 + r < win ? grow0 + r : grow0;
                const float start_x = s_sx[irow], start_y = s_sy[irow];
                lds_u8* wrow = wbase + (g * RB + r) * kPitch;
                // interior AND every coordinate a multiple of 2^-32: the fixed-point walk; anything else (a row across the frame
                // border; |cos|, |sin| or a row start below 2^-9, 0.3 % of the keypoints) takes the double-precision walk with its
                // per-batch interior test (wave-uniform choice)
                bool use_fix = fix_all;
                if (!fix_all) {
                    bad = bad || !(exact_start(start_x) && exact_start(start_y));
                    // a row whose two ends lie in the frame interior is interior throughout (the walk is a straight line)
                    const double sxd = (double)start_x, syd = (double)start_y, jl = (double)(win - 1);
                    const double pxe = __builtin_fma(jl, dcos, sxd), pye = __builtin_fma(jl, ndsin, syd);
                    const bool row_in = (unsigned)(int)floor(sxd) < (unsigned)ncols1 && (unsigned)(int)floor(syd) < (unsigned)nrows1 &&
                                        (unsigned)(int)floor(pxe) < (unsigned)ncols1 && (unsigned)(int)floor(pye) < (unsigned)nrows1;
                    use_fix = __ballot(!(row_in && fix_ok && exact32(start_x) && exact32(start_y))) == 0ull;
                }
                auto sample_rows_fix = [&]() {
                    const uint64_t X0 = fix32_from_float(start_x), Y0 = fix32_from_float(start_y);  // column 0 of this lane's row
                    uint64_t X = X0 + fcx, Y = Y0 + fcy;
                    int jb = 0;
                    if constexpr (UMAX >= 8) {
                        for (; jb + 8 * CB <= win; jb += 8 * CB) sample_batch_fix<8, CB>(quad, qs8, X, Y, fdx, fdy, wrow + jb + c);
                        if (jb + 4 * CB <= win) {
                            sample_batch_fix<4, CB>(quad, qs8, X, Y, fdx, fdy, wrow + jb + c);
                            jb += 4 * CB;
                        }
                    } else {
                        for (; jb + 4 * CB <= win; jb += 4 * CB) sample_batch_fix<4, CB>(quad, qs8, X, Y, fdx, fdy, wrow + jb + c);
                    }
                    if (jb + 2 * CB <= win) {
                        sample_batch_fix<2, CB>(quad, qs8, X, Y, fdx, fdy, wrow + jb + c);
                        jb += 2 * CB;
                    }
                    if (jb + CB <= win) {
                        sample_batch_fix<1, CB>(quad, qs8, X, Y, fdx, fdy, wrow + jb + c);
                        jb += CB;
                    }
                    if (jb < win) {  // last, partial group of columns: lanes past the window repeat its last column
                        const bool past = jb + c > win - 1;
                        uint64_t Xl = past ? add_u64(X0, fendx) : X, Yl = past ? add_u64(Y0, fendy) : Y;
                        sample_batch_fix<1, CB>(quad, qs8, Xl, Yl, fdx, fdy, wrow + min(jb + c, win - 1));
                    }
                };
                auto sample_rows = [&]() {
                    const double sxd = (double)start_x, syd = (double)start_y;
                    int jb = 0;
#define VO_BATCH(UU)                                                                                                          \
    {                                                                                                                         \
        const double jd = (double)(jb + c);                                                                                   \
        sample_batch<UU, CB>(img, quad, W, qs8, ncols1, nrows1, __builtin_fma(jd, dcos, sxd), __builtin_fma(jd, ndsin, syd), dxs,     \
                                 dys, wrow + jb + c);                                                                         \
        jb += UU * CB;                                                                                                        \
    }
                    if constexpr (UMAX >= 8) {
                        while (jb + 8 * CB <= win) VO_BATCH(8)
                        if (jb + 4 * CB <= win) VO_BATCH(4)
                    } else {
                        while (jb + 4 * CB <= win) VO_BATCH(4)
                    }
                    if (jb + 2 * CB <= win) VO_BATCH(2)
                    if (jb + CB <= win) VO_BATCH(1)
#undef VO_BATCH
                    if (jb < win) {  // last, partial group of columns: lanes past the window repeat its last column
                        const int jc = min(jb + c, win - 1);
                        const double jd = (double)jc;
                        sample_batch<1, CB>(img, quad, W, qs8, ncols1, nrows1, __builtin_fma(jd, dcos, sxd),
                                            __builtin_fma(jd, ndsin, syd), dxs, dys, wrow + jc);
                    }
                };
                if (use_fix)
                    sample_rows_fix();
                else
                    sample_rows();
              }
                wave_sync_lds();
                // INTER_AREA row pass: one (row, destination cell) per lane, the cell's pixels in table order
                const int nrw = min(WR, win - wrow0);
                if (area_fast) {
                    for (int it = lane; it < nrw * (kPatch + 1); it += 64) {
                        const int rr = it / (kPatch + 1), cell = it - rr * (kPatch + 1);
                        const lds_u8* p = wbase + rr * kPitch;
                        int tot = 0;
                        for (int q = cell * iscale; q < (cell + 1) * iscale; q++) tot += p[q];
                        s_rowi[(sidx * RS + wave * WR + rr) * (kPatch + 1) + cell] = tot;
                    }
                } else if (lane < 3 * (kPatch + 1)) {
                    const CellTab myct = s_cell[mycell];
                    // the cell's byte stream starts at pixel sx1 - 1 = stream byte 0 (the head); bytes 1 .. n are the full-weight
                    // pixels.  Weights of stream bytes 0 .. 3 per lane; +0 where the byte is not part of the cell (x + 0 == x: every
                    // partial sum is >= +0)
                    // (windows above 128 px: every cell has at least five full-weight pixels, bytes 1 .. 3 always weigh a_mid)
                    // (weights carried at 2^127: a pixel byte enters its product as the denormal float whose bits it is,
                    // mul_byte_denorm; the weights are in [2^-16, 1], so every product b * a * 2^-22 is normal and rounds as
                    // b * a does, the sums scale with it, and the result is multiplied by 2^22 before it is stored)
                    const int rp_s0 = myct.sx1 - 1, rp_sx2 = myct.sx2;
                    const float rp_am = myct.a_mid * 0x1p127f;
                    const float rp_w0 = myct.has_head ? myct.a_head * 0x1p127f : 0.f, rp_wt = myct.has_tail ? myct.a_tail * 0x1p127f : 0.f;
                    float rp_w1 = rp_am, rp_w2 = rp_am, rp_w3 = rp_am;
                    if (kRpSmall) {
                        const int n = rp_sx2 - rp_s0 - 1;
                        rp_w1 = n >= 1 ? rp_am : 0.f, rp_w2 = n >= 2 ? rp_am : 0.f, rp_w3 = n >= 3 ? rp_am : 0.f;
                    }
                    const int rp_nmin = __builtin_amdgcn_readfirstlane(myct.pad) & 0xffff, rp_nmax = __builtin_amdgcn_readfirstlane(myct.pad) >> 16;
                    // Stream bytes up to rp_nmin are full-weight pixels in every lane, and for every window size at most ONE more group
                    // of four bytes holds pixels of some lanes (vo_build_window_tables checks it).  Its weights per lane - a_mid, or +0
                    // past the lane's own count - are set up once per strip, and the group runs like the others (a compare and a
                    // select per byte were 8 of a row's ~45 instructions in the <= 128 px class).
                    const int rp_jm = max(4, 4 * ((rp_nmin + 1) >> 2));  // first stream byte of that group
                    const bool rp_m1 = rp_jm <= rp_nmax;
                    float rp_wm[4];
                    {
                        const int n = rp_sx2 - rp_s0 - 1;
#pragma unroll
                        for (int u = 0; u < 4; u++) rp_wm[u] = rp_jm + u <= n ? rp_am : 0.f;
                    }
                    // lane = (row mod 3, cell).  The pixels a
                    // cell reads are contiguous - head (sx1 - 1), full-weight pixels [sx1, sx2), tail (sx2) - so they come in
                    // as a byte stream of ALIGNED words shifted into place (v_alignbyte): one LDS read per four pixels.  The
                    // adds stay in pixel order; a pixel past a lane's own count is added with weight 0 or not at all.
                    // (running LDS pointers: an index product per row compiles to a quarter-rate 64-bit multiply-add)
                    const lds_u8* p = wbase + myrow * kPitch;
                    lds_f32* outp = (lds_f32*)s_rowv + ((sidx * RS + wave * WR + myrow) * (kPatch + 1) + mycell);
                    for (int rr = myrow; rr < nrw; rr += 3, p += 3 * kPitch, outp += 3 * (kPatch + 1)) {
                        const lds_u32* q = reinterpret_cast<const lds_u32*>(p + (rp_s0 & ~3));
                        const uint32_t rp_sh = (uint32_t)rp_s0;  // v_alignbyte uses bits [1:0]
                        uint32_t lo = q[0], hi = q[1];
                        const int pt = p[rp_sx2];
                        uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, rp_sh);
                        float acc = mul_byte_denorm<0>(w, rp_w0);  // head, or +0
                        acc += mul_byte_denorm<1>(w, rp_w1);
                        acc += mul_byte_denorm<2>(w, rp_w2);
                        acc += mul_byte_denorm<3>(w, rp_w3);
                        int j = 4, qi = 2;
                        for (; j + 3 <= rp_nmin; j += 4, qi++) {  // stream bytes j .. j + 3 are full-weight pixels in every lane
                            lo = hi;
                            hi = q[qi];
                            w = __builtin_amdgcn_alignbyte(hi, lo, rp_sh);
                            acc += mul_byte_denorm<0>(w, rp_am);
                            acc += mul_byte_denorm<1>(w, rp_am);
                            acc += mul_byte_denorm<2>(w, rp_am);
                            acc += mul_byte_denorm<3>(w, rp_am);
                        }
                        if (rp_m1) {  // j == rp_jm here; bytes past a lane's own count weigh +0
                            lo = hi;
                            hi = q[qi];
                            w = __builtin_amdgcn_alignbyte(hi, lo, rp_sh);
                            acc += mul_byte_denorm<0>(w, rp_wm[0]);
                            acc += mul_byte_denorm<1>(w, rp_wm[1]);
                            acc += mul_byte_denorm<2>(w, rp_wm[2]);
                            acc += mul_byte_denorm<3>(w, rp_wm[3]);
                        }
                        acc += mul_byte_denorm<0>((uint32_t)pt, rp_wt);  // tail, or +0
                        *outp = acc * 0x1p22f;
                    }
                }
            }
            if (NW > 1)
                __syncthreads();  // strip complete (the other buffer half is free: its fold ended before this barrier)
            else
                wave_sync_lds();
            if (area_fast) {
                if (tid <= kPatch) {
                    const int nrows = min(RS, win - i0);
                    const int ob = sidx * RS * (kPatch + 1) + tid;
                    for (int q = 0; q < nrows; q++) {
                        viacc += s_rowi[ob + q * (kPatch + 1)];
                        if (--left == 0) {
                            int out;
                            if (iscale == 1)
                                out = viacc;
                            else if (iscale == 2)
                                out = (viacc + 2) >> 2;
                            else
                                out = saturate_u8(d_cvRoundF((float)viacc * (1.f / (float)(iscale * iscale))));
                            s_patch[dy * (kPatch + 1) + tid] = (uint8_t)out;
                            dy++;
                            viacc = 0;
                            left = iscale;
                        }
                    }
                }
            } else if (tid < 3 * (kPatch + 1)) {
                const int f_dx = tid - (tid / (kPatch + 1)) * (kPatch + 1);
                int f_first = 0, f_last = -1, f_sx2 = 0, f_tail = 0;
                float f_wfirst = 0.f, f_amid = 0.f, f_atail = 0.f;
                auto fold_load = [&]() {  // source rows and weights of destination row f_d
                    if (f_d <= kPatch) {
                        const CellTab ct = s_cell[f_d];
                        f_first = ct.has_head ? ct.sx1 - 1 : ct.sx1;
                        f_last = ct.has_tail ? ct.sx2 : ct.sx2 - 1;
                        f_sx2 = ct.sx2;
                        f_tail = ct.has_tail;
                        f_wfirst = ct.has_head ? ct.a_head : (ct.sx1 < ct.sx2 ? ct.a_mid : ct.a_tail);
                        f_amid = ct.a_mid;
                        f_atail = ct.a_tail;
                    } else {
                        f_first = 1 << 30;  // never starts
                    }
                };
                fold_load();
                const int r_end = min(i0 + RS, win);  // source rows [i0, r_end) are in this strip's buffer
                // row q of the strip buffer at byte vcol + 84 q.  (Index products written as 24-bit multiplies and a running pointer: left
                // to itself the compiler turns every `vcol[q * 21]` into a quarter-rate v_mad_u64_u32, eight per destination row.)
                constexpr int kRowB = (kPatch + 1) * 4;
                const lds_u8* vcol = (const lds_u8*)((const lds_f32*)s_rowv + (sidx * RS * (kPatch + 1) + f_dx)) - i0 * kRowB;
                auto vrow = [&](int q) { return (const lds_f32*)(vcol + __mul24(q, kRowB)); };
                while (f_first < r_end) {
                    int q = max(f_first, i0);
                    const lds_f32* pq = vrow(q);
                    float acc = f_carry;
                    if (q == f_first) {  // the row begins here: OpenCV assigns beta * buf to the open sum
                        acc = f_wfirst * pq[0];
                        q++;
                        pq += kPatch + 1;
                    }
                    const int qm = min(f_sx2, r_end);  // full-weight rows [q, qm), four reads in flight, adds in row order
                    for (; q < qm; q += 4, pq += 4 * (kPatch + 1)) {
                        // (rows at and past qm are read too - at most three rows past the strip's last, inside the workgroup's LDS -
                        // and not added)
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; u++) v[u] = pq[u * (kPatch + 1)];
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const float a = acc + f_amid * v[u];
                            acc = q + u < qm ? a : acc;
                        }
                    }
                    q = max(min(q, qm), min(f_first + 1, r_end));  // the row after the last one added
                    if (f_tail && q == f_sx2 && q < r_end) {
                        acc += f_atail * *vrow(q);
                        q++;
                    }
                    if (q > f_last) {  // destination row complete
                        s_patch[__mul24(f_d, kPatch + 1) + f_dx] = (uint8_t)saturate_u8(d_cvRoundF(acc));
                        f_d += 3;
                        f_carry = 0.f;
                        fold_load();
                    } else {  // continues in the next strip
                        f_carry = acc;
                        break;
                    }
                }
            }
        }
        if (tid == 0) {  // look-ahead stage 3: the next keypoint's record (every thread has read s_q: barriers passed)
            const volatile float* kq = reinterpret_cast<const volatile float*>(kp_base + n_code);
            const float nx = kq[0], ny = kq[1], nsz = kq[2], nan_ = kq[3];
            const volatile float* dq = reinterpret_cast<const volatile float*>(dir_sc + n_code);
            const float nsin = dq[0], ncos = dq[1];
            s_q[6] = __float_as_int(nsin);
            s_q[7] = __float_as_int(ncos);
            s_q[0] = n_item;
            s_q[1] = n_code;
            s_q[2] = __float_as_int(nx);
            s_q[3] = __float_as_int(ny);
            s_q[4] = __float_as_int(nsz);
            s_q[5] = __float_as_int(nan_);
        }
        if (__syncthreads_or(bad ? 1 : 0)) {
            // coordinates are not exactly representable in closed form: sequential walk (k_describe, list 4)
            if (tid == 0) seq_list[atomicAdd(&cls_cnt[4], 1)] = code;
            continue;
        }
        describe_patch_to_vec(tid, NT, s_patch, s_grad, s_vec, desc_base + ((size_t)frame * cap + k) * 64);
    }
}

// Window class of every keypoint -> per-class work lists, counting-sorted by (frame, 256-px tile).  Keypoints that
// are close in a list are close in the frame, and the descriptor kernels hand contiguous list segments to the
// workgroups of one XCD: the windows those workgroups sample overlap, so the footprint image is served from that
// XCD's L2 instead of being re-fetched over the fabric for every keypoint (the order inside a list is otherwise
// irrelevant: every keypoint's result is independent of it).
__device__ __forceinline__ int classify_key(const vo_keypoint& kp, int frame, int W, int nbins, int* cls) {
    const float s = kp.size * 1.2f / 9.0f;
    const int win = (int)((kPatch + 1) * s);
    // (class 4: windows beyond the tiled kernels' tables - they exist only with more than three layers per octave - go straight
    // to the sequential kernel's list)
    *cls = win <= 64 ? 0 : (win <= kSmallWin ? 1 : (win <= kMidWin ? 2 : (win <= kBigWin ? 3 : 4)));
    const int bx = min(max((int)kp.x, 0), W - 1) >> 8, by = max((int)kp.y, 0) >> 8;
    const int bins_x = (W + 255) >> 8;
    return frame * nbins + min(by * bins_x + bx, nbins - 1);
}

__global__ __launch_bounds__(256) void k_classify_count(const vo_keypoint* __restrict__ kp_base, int cap,
                                                        const FrameCounters* __restrict__ cnt, int W, int nbins, int nkeys,
                                                        int* __restrict__ hist, float2* __restrict__ dir_sc) {
    const int frame = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(cnt[frame].n_raw, cap)) return;
    int cls;
    const vo_keypoint kp = kp_base[(size_t)frame * cap + k];
    const int key = classify_key(kp, frame, W, nbins, &cls);
    if (cls < 4) atomicAdd(&hist[cls * nkeys + key], 1);
    // sin / cos of the descriptor direction, one keypoint per LANE here instead of one per wavefront in the descriptor kernels
    dir_sc[(size_t)frame * cap + k] = direction_sincos(kp.angle * (float)(3.1415926535897932384626433832795 / 180));
}

// one workgroup per class: exclusive scan of the histogram in place (it becomes the scatter cursor), total -> cls_cnt
__global__ __launch_bounds__(1024) void k_classify_scan(int nkeys, int* __restrict__ hist, int* __restrict__ cls_cnt) {
    __shared__ int s_part[1024];
    int* h = hist + (size_t)blockIdx.x * nkeys;
    const int tid = threadIdx.x;
    const int per = (nkeys + 1023) / 1024;
    const int lo = min(tid * per, nkeys), hi = min(lo + per, nkeys);
    int tot = 0;
    for (int i = lo; i < hi; i++) tot += h[i];
    s_part[tid] = tot;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = tid >= d ? s_part[tid - d] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - tot;
    for (int i = lo; i < hi; i++) {
        const int c = h[i];
        h[i] = run;
        run += c;
    }
    if (tid == 1023) cls_cnt[blockIdx.x] = s_part[1023];
}

__global__ __launch_bounds__(256) void k_classify_scatter(const vo_keypoint* __restrict__ kp_base, int cap,
                                                          const FrameCounters* __restrict__ cnt, int W, int nbins, int nkeys,
                                                          int* __restrict__ hist, int* __restrict__ lists, size_t list_stride,
                                                          int* __restrict__ cls_cnt) {
    const int frame = blockIdx.y;
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= min(cnt[frame].n_raw, cap)) return;
    int cls;
    const int key = classify_key(kp_base[(size_t)frame * cap + k], frame, W, nbins, &cls);
    if (cls < 4)
        lists[cls * list_stride + atomicAdd(&hist[cls * nkeys + key], 1)] = frame * cap + k;
    else
        lists[4 * list_stride + atomicAdd(&cls_cnt[4], 1)] = frame * cap + k;
}

// ------------------------------------------------------------------------------------------------
// host: whole detect (+describe) pipeline for a frame set
// ------------------------------------------------------------------------------------------------
// a2 + a3 of the product path: the two fused Hessian + NMS launches.  dbg_det (inspection only, otherwise null) receives the
// det planes of frame 0 as the NMS reads them.
int vo_enqueue_hessian_nms(vo_ctx* ctx, FrameSet& fs, bool describe, float* dbg_det) {
    if (ctx->prm.n_octave_layers != kLayersPerOctave - 2) return enqueue_hessian_nms_generic(ctx, fs, describe, 0);
    DetectGeom g;
    g.H = fs.H;
    g.W = fs.W;
    int base = 0;
    for (int o = 0; o < kNumOctaves; o++) {
        const int lr = fs.H >> o, lc = fs.W >> o;
        g.tiles_x[o] = (lc + kTX - 1) / kTX;
        g.tiles_y[o] = (lr + kTY - 1) / kTY;
        g.tile_base[o] = base;
        base += g.tiles_x[o] * g.tiles_y[o];
    }
    g.tile_base[kNumOctaves] = base;
    {
        // octaves 0 and 1 (94 % of the samples): integral patch staged in LDS.  Octaves 2-3: taps straight from L1/L2
        // (a patch would be 2-4x the tile's own taps); their few long workgroups ride in the octave-0 grid.
        DetectGeom g2 = g;
        g2.tiles_x[0] = g2.tiles_y[0] = g2.tiles_x[1] = g2.tiles_y[1] = 0;
        // fewer than the reference's four octaves (vo_params.n_octaves): the octaves are independent, the dropped ones are not launched
        const int n_oct = std::min(ctx->prm.n_octaves, kNumOctaves);
        for (int o = n_oct; o < kNumOctaves; o++) g2.tiles_x[o] = g2.tiles_y[o] = 0;
        // keypoints the describer would delete are dropped here: 1 = the reference's two tests, 2 = upright (the first test only)
        const int filter_mode = describe ? (ctx->prm.upright ? 2 : 1) : 0;
        int base2 = 0;
        for (int o = 0; o < kNumOctaves; o++) {
            g2.tile_base[o] = base2;
            base2 += g2.tiles_x[o] * g2.tiles_y[o];
        }
        g2.tile_base[kNumOctaves] = base2;
        constexpr int T0 = 32, T1 = 24;
        const int tx1 = ((fs.W >> 1) + T1 - 3) / (T1 - 2), ty1 = ((fs.H >> 1) + T1 - 3) / (T1 - 2);
        // the two launches are independent (they append to the same keypoint list through the frame's counter): octave 1 runs
        // on the auxiliary stream beside the octave-0 grid unless the stage timers are on
        const bool beside = !ctx->profiling;
        hipStream_t s_oct1 = beside ? ctx->stream_aux : ctx->stream;
        if (beside) {
            VO_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
            VO_HIP(hipStreamWaitEvent(ctx->stream_aux, ctx->ev_fork, 0));
        }
        StageScope* sc1 = new StageScope(ctx, ST_HESSIAN_OCT1);
        if (n_oct >= 2)
            hipLaunchKernelGGL((k_hessian_nms_lds<T1, 1, 512>), dim3(tx1 * ty1, fs.F), dim3(512), 0, s_oct1, fs.sum, fs.sum_stride(),
                           fs.H, fs.W, tx1, ctx->prm.hessian_threshold, fs.kp_raw, fs.cap, fs.cnt, filter_mode, dbg_det);
        delete sc1;
        {
            StageScope sc0(ctx, ST_HESSIAN_OCT0);
            const int tx0 = (fs.W + T0 - 3) / (T0 - 2), ty0 = (fs.H + T0 - 3) / (T0 - 2);
            hipLaunchKernelGGL((k_hessian_nms_fused<T0, 256>), dim3(base2 + tx0 * ty0, fs.F), dim3(256), 0, ctx->stream, fs.sum,
                               fs.sum_stride(), fs.sum4, fs.sum4_pitch(), fs.sum4_plane(), fs.H, fs.W, tx0, tx0 * ty0, g2, ctx->prm.hessian_threshold, fs.kp_raw, fs.cap, fs.cnt,
                               filter_mode, dbg_det);
        }
        if (beside) {
            VO_HIP(hipEventRecord(ctx->ev_join, ctx->stream_aux));
            VO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
        }
        VO_HIP(hipGetLastError());
    }
    // a fifth octave (vo_params.n_octaves = 5; the reference has 4, camera_utility.hpp:35): the plain kernels after the tuned
    // launches of octaves 0 - 3, on the main stream; they append to the same lists
    if (ctx->prm.n_octaves > kNumOctaves) VO_TRY(enqueue_hessian_nms_generic(ctx, fs, describe, kNumOctaves));
    return VO_OK;
}

int vo_enqueue_detect(vo_ctx* ctx, FrameSet& fs, bool describe, bool integral_beside, hipEvent_t img_ready) {
    // integral_beside (batch runs): the integral pass - a streaming pass that needs nothing but the frames - goes to the
    // auxiliary stream, which idles through the second half of the PREVIOUS run's description (its one launch there ends
    // early), so it runs under that run's VALU-bound descriptor kernels instead of after them.  Nothing of the previous run
    // reads the integral image by then: orientation and NMS are over before the auxiliary stream's descriptor launch starts,
    // and the sequential fallback takes its directions from the keypoint records.  img_ready: the frames' upload (may be null).
    if (integral_beside && describe && !ctx->profiling) {
        if (img_ready) VO_HIP(hipStreamWaitEvent(ctx->stream_aux, img_ready, 0));
        VO_TRY(vo_enqueue_integral(ctx, fs, ctx->stream_aux));
        VO_HIP(hipEventRecord(ctx->ev_integral, ctx->stream_aux));
        VO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_integral, 0));
    } else {
        if (img_ready) VO_HIP(hipStreamWaitEvent(ctx->stream, img_ready, 0));
        VO_TRY(vo_enqueue_integral(ctx, fs, ctx->stream));
    }
    VO_TRY(vo_enqueue_hessian_nms(ctx, fs, describe, nullptr));
    // The footprint image of the descriptor sampler depends on the frame only: it is built on the auxiliary stream beside the
    // sort (one workgroup per frame: half the CUs idle) and the orientation pass (VALU-bound, while this is a streaming pass).
    // With the stage timers on it stays in its stage on the main stream.
    const bool quads_beside = describe && !ctx->profiling;
    auto launch_quads = [&](hipStream_t st) {
        if (fs.W % 4 == 0 && fs.img_stride() % 4 == 0)
            hipLaunchKernelGGL(k_quads4, dim3(((fs.W + 7) / 8 + 31) / 32, (fs.H + 3) / 4, fs.F), dim3(256), 0, st, fs.img, fs.img_stride(), fs.H, fs.W,
                               fs.quad);
        else
            hipLaunchKernelGGL(k_quads, dim3((fs.W + 255) / 256, fs.H, fs.F), dim3(256), 0, st, fs.img, fs.img_stride(), fs.H, fs.W, fs.quad);
    };
    if (quads_beside) {
        VO_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
        VO_HIP(hipStreamWaitEvent(ctx->stream_aux, ctx->ev_fork, 0));
        launch_quads(ctx->stream_aux);
        VO_HIP(hipEventRecord(ctx->ev_quads, ctx->stream_aux));
    }
    {
        StageScope sc(ctx, ST_SORT);
        hipLaunchKernelGGL(k_count_clamp, dim3((fs.F + 255) / 256), dim3(256), 0, ctx->stream, fs.cnt, fs.cap, fs.F);
        // a few frames (a single pair): the rank sort spreads over the chip, the bitonic network is one workgroup per frame
        if (fs.cap <= kBitonicMax && fs.F >= 16) {
            int np = 1024;
            while (np < fs.cap) np <<= 1;
            const size_t lds = (size_t)np * sizeof(unsigned long long);
            hipLaunchKernelGGL(k_sort_bitonic, dim3(fs.F), dim3(1024), lds, ctx->stream, fs.kp_raw, fs.cap, fs.cnt, fs.kp);
        } else {
            dim3 gr((fs.cap + 255) / 256, (fs.cap + kRankChunk - 1) / kRankChunk, fs.F);
            hipLaunchKernelGGL(k_rank, gr, dim3(256), 0, ctx->stream, fs.kp_raw, fs.cap, fs.cnt, fs.rank);
            hipLaunchKernelGGL(k_scatter, dim3((fs.cap + 255) / 256, fs.F), dim3(256), 0, ctx->stream, fs.kp_raw, fs.cap,
                               fs.cnt, fs.rank, fs.kp);
        }
        VO_HIP(hipGetLastError());
    }
    {
        float* d = describe ? fs.desc : nullptr;
        if (describe) {
            const size_t ls = (size_t)fs.F * fs.cap;
            {
                StageScope sc(ctx, ST_ORIENTATION);
                if (ctx->prm.upright)
                    hipLaunchKernelGGL(k_upright_angle, dim3((fs.cap + 255) / 256, fs.F), dim3(256), 0, ctx->stream, fs.kp, fs.cap, fs.cnt);
                else
                    hipLaunchKernelGGL(k_orientation, dim3((fs.cap + 7) / 8, fs.F), dim3(256), 0, ctx->stream, fs.sum, fs.sum_stride(),
                                       fs.H, fs.W, fs.kp, fs.cap, fs.cnt);
            }
            {
                StageScope sc(ctx, ST_DESC_PREP);
                const int nbins = fs.nbins(), nkeys = fs.F * nbins;
                const dim3 gk((fs.cap + 255) / 256, fs.F);
                hipLaunchKernelGGL(k_classify_count, gk, dim3(256), 0, ctx->stream, fs.kp, fs.cap, fs.cnt, fs.W, nbins, nkeys,
                                   fs.cls_hist, fs.dir_sc);
                hipLaunchKernelGGL(k_classify_scan, dim3(4), dim3(1024), 0, ctx->stream, nkeys, fs.cls_hist, fs.cls_cnt);
                hipLaunchKernelGGL(k_classify_scatter, gk, dim3(256), 0, ctx->stream, fs.kp, fs.cap, fs.cnt, fs.W, nbins, nkeys,
                                   fs.cls_hist, fs.cls_list, ls, fs.cls_cnt);
                if (quads_beside)
                    VO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_quads, 0));
                else
                    launch_quads(ctx->stream);
            }
            // upright windows are pixel rectangles: every keypoint takes the sequential kernel, which has that branch (the flag is off
            // in the reference, camera_utility.hpp:38: correctness only, no tiled form)
            // (likewise the 128-float descriptors of `extended`, :37: the sequential kernel has that form of the last step)
            const int force_seq = (ctx->force_seq_describe || ctx->prm.upright || ctx->prm.extended) ? 1 : 0;
#define LAUNCH_TILED(STAGE, HI, NW, RB, UMAX, G, CH, CLS, GRID, STREAM)                                                             \
    {                                                                                                                              \
        StageScope sc(ctx, STAGE);                                                                                                 \
        hipLaunchKernelGGL((k_describe_tiled<HI, NW, RB, UMAX, G, CH>), dim3(GRID), dim3((NW) * 64), 0, STREAM, fs.img,               \
                           fs.img_stride(), fs.quad, fs.sum, fs.sum_stride(), fs.H, fs.W, fs.kp, d, fs.cap,                         \
                           fs.cls_list + (size_t)(CLS) * ls, fs.cls_cnt, CLS, fs.cls_list + (size_t)4 * ls, force_seq, \
                           (const CellTab*)ctx->d_win_cell, (const PixTab*)ctx->d_win_pix, fs.dir_sc);                               \
    }
            // The heaviest class goes to a stream of its own, beside the other three: its keypoints are long (0.35 ms each), and
            // the last of them would otherwise hold the whole chip while the next launch waits.  With the stage timers on, the
            // four launches stay on one stream so that each is timed alone.
            const bool beside = !ctx->profiling;
            hipStream_t s_big = beside ? ctx->stream_aux : ctx->stream;
            if (beside) {
                VO_HIP(hipEventRecord(ctx->ev_fork, ctx->stream));
                VO_HIP(hipStreamWaitEvent(ctx->stream_aux, ctx->ev_fork, 0));
            }
            LAUNCH_TILED(ST_DESC_W768, kBigWin, 4, 8, 8, 1, 1, 3, 1024, s_big);
            LAUNCH_TILED(ST_DESC_W320, kMidWin, 4, 8, 8, 1, 2, 2, 1280, ctx->stream);
            LAUNCH_TILED(ST_DESC_W128, kSmallWin, 1, 8, 8, 3, 4, 1, 5120, ctx->stream);
            LAUNCH_TILED(ST_DESC_W64, 64, 1, 8, 4, 4, 8, 0, 5120, ctx->stream);
            if (beside) {
                VO_HIP(hipEventRecord(ctx->ev_join, ctx->stream_aux));
                VO_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
            }
#undef LAUNCH_TILED
            {
                // keypoints whose window coordinates are not exact in closed form (list 4): sequential walk
                StageScope sc(ctx, ST_DESC_SEQ);
                // (more than three layers per octave: keypoint sizes up to 48 * n_octave_layers + 120, windows up to kHugeWin)
                // (upright / extended send EVERY keypoint here: two workgroups per CU instead of one per CU)
                const int seq_grid = (ctx->prm.upright || ctx->prm.extended) ? 512 : 256;
                if (ctx->prm.n_octaves > kNumOctaves)  // sizes up to 528, windows up to 1478: 150 KB of LDS
                    hipLaunchKernelGGL((k_describe<0, kOct5Win, 256>), dim3(seq_grid / 2), dim3(256), 0, ctx->stream, fs.img, fs.img_stride(),
                                       fs.sum, fs.sum_stride(), fs.H, fs.W, fs.kp, d, fs.cap, fs.cls_list + (size_t)4 * ls, fs.cls_cnt, 4,
                                       ctx->prm.upright ? 2 : 1, ctx->prm.extended);
                else if (ctx->prm.n_octave_layers > kLayersPerOctave - 2)
                    hipLaunchKernelGGL((k_describe<0, kHugeWin, 256>), dim3(seq_grid / 2), dim3(256), 0, ctx->stream, fs.img, fs.img_stride(),
                                       fs.sum, fs.sum_stride(), fs.H, fs.W, fs.kp, d, fs.cap, fs.cls_list + (size_t)4 * ls, fs.cls_cnt, 4,
                                       ctx->prm.upright ? 2 : 1, ctx->prm.extended);
                else
                    hipLaunchKernelGGL((k_describe<0, kBigWin, 256>), dim3(seq_grid), dim3(256), 0, ctx->stream, fs.img, fs.img_stride(),
                                       fs.sum, fs.sum_stride(), fs.H, fs.W, fs.kp, d, fs.cap, fs.cls_list + (size_t)4 * ls, fs.cls_cnt, 4,
                                       ctx->prm.upright ? 2 : 1, ctx->prm.extended);
            }
            VO_HIP(hipGetLastError());
        }
    }
    return VO_OK;
}

// > 64 KB of dynamic LDS needs the opt-in, and the attribute belongs to the DEVICE's copy of the kernel: set for every context
// (vo_ctx_create, with the context's device current), not once per process
int vo_detect_init(vo_ctx* ctx) {
    (void)ctx;
    VO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_sort_bitonic), hipFuncAttributeMaxDynamicSharedMemorySize,
                               kBitonicMax * (int)sizeof(unsigned long long)));
    return VO_OK;
}

int vo_upload_detect_tables(const DetectTables& t) {
    VO_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tab), &t, sizeof(t)));
    return VO_OK;
}
