// tools/ubench/valu_rates.hip -- measures per-instruction VALU issue cost on gfx950 for the
// f64 operations the sweep kernels are made of (diagnostic, not part of the library).
// hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int OP> __global__ void k(double *out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3;
    double a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    const double b = 1.0000001, c = 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; i++) {
#define STEP(v)                                                                               \
    if (OP == 0) v = __builtin_fma(v, b, c);                                                  \
    else if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(b));               \
    else if (OP == 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(c));               \
    else if (OP == 3) asm volatile("v_rcp_f64 %0, %0" : "+v"(v));                             \
    else if (OP == 4) asm volatile("v_rsq_f64 %0, %0" : "+v"(v));                             \
    else if (OP == 5) asm volatile("v_div_fixup_f64 %0, %0, %1, %1" : "+v"(v) : "v"(b));     \
    else if (OP == 6) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v) : "v"(b));               \
    else if (OP == 7) { int lo = __double2loint(v); lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true); v = __hiloint2double(__double2hiint(v), lo); } \
    else if (OP == 8) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(v) : "v"(b) : "vcc"); \
    else if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(*(int*)&v) : "v"(3) : );     \
    else if (OP == 10) asm volatile("v_div_fmas_f64 %0, %0, %1, %1" : "+v"(v) : "v"(b));             \
    else if (OP == 11) asm volatile("v_mov_b64 %0, %1" : "=v"(v) : "v"(b));                             \
    else if (OP == 12) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v), "v"(b) : "vcc");             \
    else if (OP == 13) asm volatile("v_min_f64 %0, %0, %1" : "+v"(v) : "v"(b));                         \
    else if (OP == 14) asm volatile("v_mov_b32 %0, %1" : "=v"(*(int*)&v) : "v"(3));                     \
    else if (OP == 15) asm volatile("v_sqrt_f64 %0, %0" : "+v"(v));                                     \
    else if (OP == 16) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(v), "v"(3) : "vcc");          \
    else if (OP == 17) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(v) : "v"(b));                     \
    else if (OP == 18) asm volatile("v_add_f64 %0, |%0|, -%1" : "+v"(v) : "v"(c));                      \
    else if (OP == 19) asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(v) : "v"(b));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 12345.678) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[1] = (double)(t1 - t0); out[2] = (double)(r1 - r0); }
}

template <int OP> void run(const char *name, int waves_per_simd) {
    double *d; hipMalloc(&d, 64);
    const int iters = 20000;
    dim3 grid(256 * 4 * waves_per_simd / 4), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 100, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    const double ninstr = 16.0 * iters * waves_per_simd;     // per SIMD
    const double clk = h[1] / (h[2] / 100e6) / 1e9;          // GHz
    printf("%-16s waves/SIMD %d  %.3f ms  clock %.2f GHz  cycles/instr(per SIMD) %.2f\n", name, waves_per_simd, ms,
           clk, h[1] / ninstr);
    hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f64", w); run<1>("v_mul_f64", w); run<2>("v_add_f64", w); run<3>("v_rcp_f64", w);
        run<4>("v_rsq_f64", w); run<5>("v_div_fixup_f64", w); run<6>("v_max_f64", w); run<7>("v_mov_dpp", w);
        run<8>("v_div_scale_f64", w); run<9>("v_cndmask_b32", w); run<10>("v_div_fmas_f64", w);
        run<11>("v_mov_b64", w); run<12>("v_cmp_lt_f64", w); run<13>("v_min_f64", w); run<14>("v_mov_b32", w);
        run<15>("v_sqrt_f64", w); run<16>("v_cmp_class_f64", w); run<17>("v_fma_f64 asm", w); run<18>("v_add_f64 mods", w);
        run<19>("v_pk_mov_b32", w);
    }
    return 0;
}
