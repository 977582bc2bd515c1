// Micro-benchmark: cost of the bilinear-footprint gathers of the descriptor window on gfx950.
// One wavefront = 8 rows x 8 columns of unit-spaced samples rotated by theta; variants:
//   A: u8 image, two unaligned 16-bit loads per sample       (current kernel)
//   C: row-pair image (2 B / pixel), one 2-byte-aligned 32-bit load per sample
//   Q: quad image (4 B / pixel), one aligned 32-bit load per sample
// Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int W = 1920, H = 1080;

template <int MODE, bool SWAP>
__global__ __launch_bounds__(64) void k_gather(const uint8_t* __restrict__ img, const uint8_t* __restrict__ pairs,
                                               const uint32_t* __restrict__ quads, float c, float s, int win, int nkp,
                                               uint32_t* out) {
    const int lane = threadIdx.x;
    const int r = SWAP ? (lane & 7) : (lane >> 3), cc = SWAP ? (lane >> 3) : (lane & 7);
    uint32_t acc = 0;
    for (int kp = blockIdx.x; kp < nkp; kp += gridDim.x) {
        // pseudo-random window centre, well inside the frame
        uint32_t h = kp * 2654435761u;
        const float cx = 420.f + (h % 1080u), cy = 420.f + ((h >> 12) % 240u);
        for (int i0 = 0; i0 < win; i0 += 8) {
            for (int j0 = 0; j0 < win; j0 += 64) {
                uint32_t t[8][2];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const float fi = (float)(i0 + r - win / 2), fj = (float)(j0 + u * 8 + cc - win / 2);
                    const float px = cx + fj * c + fi * s, py = cy - fj * s + fi * c;
                    const int ix = (int)floorf(px), iy = (int)floorf(py);
                    const uint32_t off = (uint32_t)(iy * W + ix);
                    if (MODE == 0) {
                        uint16_t a, b;
                        __builtin_memcpy(&a, img + off, 2);
                        __builtin_memcpy(&b, img + off + W, 2);
                        t[u][0] = a;
                        t[u][1] = b;
                    } else if (MODE == 1) {
                        uint32_t a;
                        __builtin_memcpy(&a, pairs + 2 * off, 4);
                        t[u][0] = a;
                        t[u][1] = 0;
                    } else {
                        t[u][0] = quads[off];
                        t[u][1] = 0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) acc += t[u][0] * 3u + t[u][1];
            }
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE, bool SWAP>
float run(const uint8_t* img, const uint8_t* pairs, const uint32_t* quads, float theta, int win, int nkp, uint32_t* out) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const float c = cosf(theta), s = sinf(theta);
    k_gather<MODE, SWAP><<<4096, 64>>>(img, pairs, quads, c, s, win, nkp, out);
    hipEventRecord(a);
    k_gather<MODE, SWAP><<<4096, 64>>>(img, pairs, quads, c, s, win, nkp, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    std::vector<uint8_t> h((size_t)W * H);
    for (auto& v : h) v = rand();
    uint8_t *img, *pairs;
    uint32_t *quads, *out;
    hipMalloc(&img, (size_t)W * H + 4096);
    hipMalloc(&pairs, (size_t)2 * W * H + 4096);
    hipMalloc(&quads, (size_t)4 * W * H + 4096);
    hipMalloc(&out, 64);
    hipMemcpy(img, h.data(), h.size(), hipMemcpyHostToDevice);
    hipMemset(pairs, 1, (size_t)2 * W * H);
    hipMemset(quads, 1, (size_t)4 * W * H);
    const int nkp = 65536;
    for (int win : {64, 192}) {
        const double samples = (double)nkp * ((win + 7) / 8 * 8) * ((win + 63) / 64 * 64);
        for (float deg : {0.f, 20.f, 45.f, 70.f, 90.f}) {
            const float th = deg * 3.14159265f / 180.f;
            const float a0 = run<0, false>(img, pairs, quads, th, win, nkp, out), a1 = run<0, true>(img, pairs, quads, th, win, nkp, out);
            const float c0 = run<1, false>(img, pairs, quads, th, win, nkp, out), c1 = run<1, true>(img, pairs, quads, th, win, nkp, out);
            const float q0 = run<2, false>(img, pairs, quads, th, win, nkp, out), q1 = run<2, true>(img, pairs, quads, th, win, nkp, out);
            printf("win %3d theta %2.0f  Gsamples/s: A %.0f (swap %.0f)  C %.0f (swap %.0f)  Q %.0f (swap %.0f)\n", win, deg,
                   samples / a0 * 1e-6, samples / a1 * 1e-6, samples / c0 * 1e-6, samples / c1 * 1e-6, samples / q0 * 1e-6,
                   samples / q1 * 1e-6);
        }
    }
    return 0;
}
