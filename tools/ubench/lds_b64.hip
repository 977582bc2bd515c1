// ds_read_b64 at dword-aligned but not 8-byte-aligned addresses: rate and values, against aligned ds_read_b64 and ds_read2_b32.
// Build: hipcc --offload-arch=gfx950 -O2 -o lds_b64 lds_b64.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kIters = 2048;
typedef __attribute__((address_space(3))) uint32_t lds_u32;

template <int MODE>  // 0: aligned b64, 1: odd-dword b64, 2: read2_b32 (odd), 3: two rows of 16 lanes at pitch 65, odd; 4: same at pitch 96
__global__ __launch_bounds__(256) void k_lds(uint32_t* out) {
    __shared__ uint32_t s[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) s[i] = i * 2654435761u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int idx;
    if (MODE == 0) idx = 2 * lane;
    else if (MODE == 1 || MODE == 2) idx = 2 * lane + 1;
    else if (MODE == 3) idx = (lane >> 4) * 65 + 2 * (lane & 15) + 1;
    else idx = (lane >> 4) * 96 + 2 * (lane & 15) + 1;
    idx += wave * 1024;
    const uint32_t addr = (uint32_t)(uintptr_t)(lds_u32*)s + idx * 4;
    uint32_t acc0 = 0, acc1 = 0;
    for (int i = 0; i < kIters; i++) {
        uint32_t a0, a1, b0, b1, c0, c1, d0, d1;
        if (MODE == 2) {
            asm volatile("ds_read2_b32 %0, %4 offset0:0 offset1:1\n ds_read2_b32 %1, %4 offset0:2 offset1:3\n"
                         "ds_read2_b32 %2, %4 offset0:4 offset1:5\n ds_read2_b32 %3, %4 offset0:6 offset1:7\n s_waitcnt lgkmcnt(0)"
                         : "=v"(*(uint64_t*)&a0), "=v"(*(uint64_t*)&b0), "=v"(*(uint64_t*)&c0), "=v"(*(uint64_t*)&d0) : "v"(addr) : "memory");
        }
        uint64_t qa, qb, qc, qd;
        if (MODE != 2) {
            asm volatile("ds_read_b64 %0, %4 offset:0\n ds_read_b64 %1, %4 offset:8\n ds_read_b64 %2, %4 offset:16\n ds_read_b64 %3, %4 offset:24\n s_waitcnt lgkmcnt(0)"
                         : "=v"(qa), "=v"(qb), "=v"(qc), "=v"(qd) : "v"(addr) : "memory");
        } else {
            qa = *(uint64_t*)&a0; qb = *(uint64_t*)&b0; qc = *(uint64_t*)&c0; qd = *(uint64_t*)&d0;
        }
        acc0 += (uint32_t)qa + (uint32_t)qb + (uint32_t)qc + (uint32_t)qd;
        acc1 += (uint32_t)(qa >> 32) + (uint32_t)(qb >> 32) + (uint32_t)(qc >> 32) + (uint32_t)(qd >> 32);
        (void)a1; (void)b1; (void)c1; (void)d1;
    }
    // expected: sum over the 8 dwords idx .. idx + 7
    uint32_t e0 = 0, e1 = 0;
    for (int k = 0; k < 8; k += 2) { e0 += s[idx + k]; e1 += s[idx + k + 1]; }
    out[blockIdx.x * 256 + threadIdx.x] = (acc0 == e0 * kIters && acc1 == e1 * kIters) ? 0u : 1u;
}

int main() {
    uint32_t* out;
    const int blocks = 256 * 4;  // 4 workgroups of 4 wavefronts per CU
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    struct { const char* name; void (*fn)(uint32_t*); } ks[] = {{"ds_read_b64 aligned", k_lds<0>}, {"ds_read_b64 odd dword", k_lds<1>}, {"ds_read2_b32 odd dword", k_lds<2>},
                                                                 {"ds_read_b64 odd, 4 rows x 16 lanes, pitch 65", k_lds<3>}, {"ds_read_b64 odd, 4 rows x 16 lanes, pitch 96", k_lds<4>}};
    static uint32_t h[blocks * 256];
    for (auto& k : ks) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; rep++) {
            CHK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out);
            CHK(hipEventRecord(e1, 0));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        CHK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
        unsigned bad = 0;
        for (unsigned v : h) bad += v;
        // per CU: 16 wavefronts * kIters * 4 reads of 512 B
        const double clk = best * 1e-3 * 2.4e9 / (16.0 * kIters * 4);
        printf("%-48s %7.3f ms  %5.2f clk per wavefront read (512 B) per CU  wrong lanes %u\n", k.name, best, clk, bad);
    }
    return 0;
}
