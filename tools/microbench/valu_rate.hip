// Micro-benchmark: issue rate of the VALU operations the descriptor sampler uses, on gfx950.
// Each kernel runs 8 independent dependency chains per lane so that latency is hidden; the result is
// wave-instructions per cycle per SIMD (1/4 = one 64-wide instruction every 4 cycles = "full rate").
// Build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

constexpr int ITERS = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void k(double* out, double seed) {
    double d[CH];
    float f[CH];
    int n[CH];
    for (int c = 0; c < CH; c++) {
        d[c] = seed + threadIdx.x * 1e-3 + c;
        f[c] = (float)d[c];
        n[c] = c;
    }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (OP == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[c]) : "v"(f[(c + 1) % CH]));
            if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (OP == 2) asm volatile("v_fract_f64 %0, %1" : "=v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[c]) : "v"(d[c]));
            if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[c]) : "v"(d[c]));
            if (OP == 5) asm volatile("v_floor_f64 %0, %1" : "=v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (OP == 6) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[c]) : "v"(d[(c + 1) % CH]));
            if (OP == 8) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[c]) : "v"(n[c]));
            if (OP == 9) asm volatile("v_rndne_f32 %0, %1" : "=v"(f[c]) : "v"(f[(c + 1) % CH]));
            if (OP == 10) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(d[c]) : "v"(n[c]));
            if (OP == 11) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(n[c]) : "v"(n[(c + 1) % CH]));
            if (OP == 12) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[c]) : "v"(f[c]));
        }
    }
    double s = 0;
    for (int c = 0; c < CH; c++) s += d[c] + f[c] + n[c];
    if (s == 123.456) out[0] = s;
}

template <int OP>
void run(const char* name, double* out) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a);
    (void)hipEventCreate(&b);
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU: 8 waves per SIMD
    k<OP><<<blocks, 256>>>(out, 1.0);
    (void)hipEventRecord(a);
    k<OP><<<blocks, 256>>>(out, 1.0);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    const double winstr = (double)blocks * 4 * ITERS * CH;  // wave instructions
    const double per_simd_per_clk = winstr / (1024.0 * ms * 1e-3 * 2.4e9);
    printf("%-18s %.3f ms  %.3f wave-instr/clk/SIMD  (%.1f clk per instr)\n", name, ms, per_simd_per_clk, 1.0 / per_simd_per_clk);
}

int main() {
    double* out;
    (void)hipMalloc(&out, 64);
    run<0>("v_add_f32", out);
    run<1>("v_add_f64", out);
    run<6>("v_fma_f64", out);
    run<2>("v_fract_f64", out);
    run<5>("v_floor_f64", out);
    run<3>("v_cvt_f32_f64", out);
    run<4>("v_cvt_i32_f64", out);
    run<10>("v_cvt_f64_i32", out);
    run<12>("v_cvt_f64_f32", out);
    run<7>("v_pk_mul_f32", out);
    run<8>("v_cvt_f32_ubyte1", out);
    run<9>("v_rndne_f32", out);
    run<11>("v_mad_i32_i24", out);
    return 0;
}
