// Relative issue cost of the VALU instructions the descriptor sampler is made of (gfx950).  Every kernel runs the same
// loop of 8 independent chains of one instruction; the time relative to v_add_f32 is the instruction's cost in issue slots.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kIters = 4096;

#define KERNEL32(name, INSTR)                                                                       \
    __global__ __launch_bounds__(256) void name(float* out) {                                        \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        for (int i = 0; i < kIters; i++) {                                                           \
            asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n" INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7) \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                 \
    }
#define KERNEL64(name, INSTR)                                                                       \
    __global__ __launch_bounds__(256) void name(float* out) {                                        \
        double r0 = threadIdx.x + 0.3, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        for (int i = 0; i < kIters; i++) {                                                           \
            asm volatile(INSTR(%0) "\n" INSTR(%1) "\n" INSTR(%2) "\n" INSTR(%3) "\n" INSTR(%4) "\n" INSTR(%5) "\n" INSTR(%6) "\n" INSTR(%7) \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)); \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7);       \
    }

#define I_ADD_F32(r) "v_add_f32 " #r ", 1.0, " #r
#define I_MUL_F32(r) "v_mul_f32 " #r ", 1.0, " #r
#define I_CVT_UB(r) "v_cvt_f32_ubyte1 " #r ", " #r
#define I_MUL24(r) "v_mul_i32_i24 " #r ", 3, " #r
#define I_ADDLSHL(r) "v_add_lshl_u32 " #r ", " #r ", " #r ", 2"
#define I_CVT_PK_U8(r) "v_cvt_pk_u8_f32 " #r ", " #r ", 0, 0"
#define I_MED3(r) "v_med3_f32 " #r ", " #r ", 1.0, 2.0"
#define I_PK_MUL(r) "v_pk_mul_f32 " #r ", " #r ", " #r
#define I_PK_ADD(r) "v_pk_add_f32 " #r ", " #r ", " #r
#define I_PK_FMA(r) "v_pk_fma_f32 " #r ", " #r ", " #r ", " #r
#define I_ADD_F64(r) "v_add_f64 " #r ", " #r ", 1.0"
#define I_FMA_F64(r) "v_fma_f64 " #r ", " #r ", 1.0, 1.0"
#define I_FRACT_F64(r) "v_fract_f64 " #r ", " #r
#define I_FLOOR_F64(r) "v_floor_f64 " #r ", " #r
// conversions between 64 and 32 bit: separate source and destination registers
#define KERNELCVT(name, OP, TS, TD)                                                                  \
    __global__ __launch_bounds__(256) void name(float* out) {                                        \
        TS s0 = (TS)(threadIdx.x + 0.3), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;                      \
        TD d0, d1, d2, d3, d4, d5, d6, d7;                                                           \
        for (int i = 0; i < kIters; i++) {                                                           \
            asm volatile(OP " %0, %8\n" OP " %1, %9\n" OP " %2, %10\n" OP " %3, %11\n" OP " %4, %8\n" OP " %5, %9\n" OP " %6, %10\n" OP " %7, %11" \
                         : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "=&v"(d4), "=&v"(d5), "=&v"(d6), "=&v"(d7)                             \
                         : "v"(s0), "v"(s1), "v"(s2), "v"(s3));                                      \
        }                                                                                            \
        out[blockIdx.x * 256 + threadIdx.x] = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);       \
    }
#define I_ADD_CO(r) "v_add_co_u32 " #r ", vcc, 3, " #r
#define I_LSHL_ADD64(r) "v_lshl_add_u64 " #r ", " #r ", 0, " #r

KERNEL32(k_add_f32, I_ADD_F32)
KERNEL32(k_mul_f32, I_MUL_F32)
KERNEL32(k_cvt_ubyte, I_CVT_UB)
KERNEL32(k_mul24, I_MUL24)
KERNEL32(k_add_lshl, I_ADDLSHL)
KERNEL32(k_cvt_pk_u8, I_CVT_PK_U8)
KERNEL32(k_med3, I_MED3)
KERNEL32(k_add_co, I_ADD_CO)
KERNEL64(k_pk_mul, I_PK_MUL)
KERNEL64(k_pk_add, I_PK_ADD)
KERNEL64(k_pk_fma, I_PK_FMA)
KERNEL64(k_add_f64, I_ADD_F64)
KERNEL64(k_fma_f64, I_FMA_F64)
KERNEL64(k_fract_f64, I_FRACT_F64)
KERNEL64(k_floor_f64, I_FLOOR_F64)
KERNELCVT(k_cvt_f32_f64, "v_cvt_f32_f64", double, float)
KERNELCVT(k_cvt_i32_f64, "v_cvt_i32_f64", double, int)
KERNELCVT(k_cvt_f64_i32, "v_cvt_f64_i32", int, double)
KERNEL64(k_lshl_add_u64, I_LSHL_ADD64)
#define I_CVT_F32_U32(r) "v_cvt_f32_u32 " #r ", " #r
#define I_SUB_LIT(r) "v_sub_f32 " #r ", 0x4f800000, " #r
#define I_MUL_LIT(r) "v_mul_f32 " #r ", 0x1f800000, " #r
#define I_SUB_INL(r) "v_sub_f32 " #r ", 1.0, " #r
#define I_SUB_SGPR(r) "v_sub_f32 " #r ", s20, " #r
KERNEL32(k_cvt_f32_u32, I_CVT_F32_U32)
KERNEL32(k_sub_lit, I_SUB_LIT)
KERNEL32(k_mul_lit, I_MUL_LIT)
KERNEL32(k_sub_inl, I_SUB_INL)
KERNEL32(k_sub_sgpr, I_SUB_SGPR)

int main() {
    float* out;
    const int blocks = 256 * 8;  // 8 workgroups of 4 wavefronts per CU: 8 wavefronts per SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * sizeof(float)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    struct { const char* name; void (*fn)(float*); } ks[] = {
        {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32}, {"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_mul_i32_i24", k_mul24},
        {"v_add_lshl_u32", k_add_lshl}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8}, {"v_med3_f32", k_med3}, {"v_add_co_u32", k_add_co},
        {"v_pk_mul_f32", k_pk_mul}, {"v_pk_add_f32", k_pk_add}, {"v_pk_fma_f32", k_pk_fma}, {"v_add_f64", k_add_f64},
        {"v_fma_f64", k_fma_f64}, {"v_fract_f64", k_fract_f64}, {"v_floor_f64", k_floor_f64}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_cvt_i32_f64", k_cvt_i32_f64}, {"v_cvt_f64_i32", k_cvt_f64_i32}, {"v_lshl_add_u64", k_lshl_add_u64},
        {"v_cvt_f32_u32", k_cvt_f32_u32}, {"v_sub_f32 literal", k_sub_lit}, {"v_mul_f32 literal", k_mul_lit}, {"v_sub_f32 inline", k_sub_inl},
        {"v_sub_f32 sgpr", k_sub_sgpr}};
    double base = 0;
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
        // wave-instructions per SIMD: 8 waves * kIters * 8
        const double per_instr_ns = best * 1e6 / (8.0 * kIters * 8);
        if (base == 0) base = per_instr_ns;
        printf("%-18s %7.3f ms  %6.3f ns per wavefront instruction per SIMD  = %5.2f x v_add_f32\n", k.name, best, per_instr_ns, per_instr_ns / base);
    }
    return 0;
}
