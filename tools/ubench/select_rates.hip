// tools/ubench/select_rates.hip -- what does a per-lane select cost on gfx950?  (diagnostic, not part of the library)
// v_cndmask_b32 measured ~5x an f64 add in valu_rates.hip; this isolates why and prices the alternatives.
// hipcc --offload-arch=gfx950 -O3 -o select_rates select_rates.hip && ./select_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP> __global__ void k(double *out, int iters, double seed) {
    double a0 = seed + threadIdx.x, a1 = a0 * 1.1, a2 = a0 * 1.2, a3 = a0 * 1.3;
    double a4 = a0 * 1.4, a5 = a0 * 1.5, a6 = a0 * 1.6, a7 = a0 * 1.7;
    const double b = 1.0000001, c = 1e-9;
    int three = 3;
    asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a0), "v"(b) : "vcc");
    asm volatile("s_mov_b64 s[20:21], vcc" : : : "s20", "s21");
    for (int i = 0; i < iters; i++) {
#define LO(v) (*(int *)&v)
#define STEP(v)                                                                                              \
    if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(c));                                     \
    else if (OP == 1) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(LO(v)) : "v"(three));               \
    else if (OP == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(LO(v)) : "v"(three));      \
    else if (OP == 3) { asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v), "v"(b) : "vcc"); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(LO(v)) : "v"(three)); } \
    else if (OP == 4) asm volatile("v_mov_b32 %0, %1" : "+v"(LO(v)) : "v"(three));                            \
    else if (OP == 5) asm volatile("v_and_b32 %0, %0, %1" : "+v"(LO(v)) : "v"(three));                        \
    else if (OP == 6) asm volatile("v_bfi_b32 %0, %1, %0, %1" : "+v"(LO(v)) : "v"(three));                    \
    else if (OP == 7) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(v) : "v"(c)); asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(three) : "v"(LO(a7))); } \
    else if (OP == 8) asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n s_nop 0" : "+v"(LO(v)) : "v"(three));     \
    else if (OP == 9) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v) : "v"(b));                                \
    else if (OP == 10) asm volatile("v_add_u32 %0, %0, %1" : "+v"(LO(v)) : "v"(three));                       \
    else if (OP == 11) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(v), "v"(b) : "vcc");                   \
    else if (OP == 12) asm volatile("v_cmp_lt_f64 s[22:23], %0, %1" : : "v"(v), "v"(b) : "s22", "s23");       \
    else if (OP == 13) asm volatile("v_cndmask_b32 %0, %1, %1, vcc" : "=v"(LO(v)) : "v"(three));             \
    else if (OP == 14) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v) : "v"(b));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
    }
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + three;
    if (s == 12345.678) out[0] = s;
}

template <int OP> void run(const char *name, int w, float base) {
    double *d; (void)hipMalloc(&d, 64);
    const int iters = 20000;
    dim3 grid(256 * w), block(256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, 100, 1.5);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, grid, block, 0, 0, d, iters, 1.5);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    // ns per STEP per SIMD: w waves per SIMD each issue 16*iters STEPs
    printf("%-44s waves/SIMD %d  %7.3f ms  %6.2f ns per STEP per SIMD\n", name, w, ms, ms * 1e6 / (16.0 * iters * w));
    (void)hipFree(d);
}

int main() {
    for (int w : {1, 2, 4}) {
        run<0>("v_add_f64", w, 0); run<14>("v_mul_f64", w, 0); run<9>("v_max_f64", w, 0);
        run<1>("v_cndmask_b32 (vcc)", w, 0); run<2>("v_cndmask_b32_e64 (sgpr pair)", w, 0);
        run<13>("v_cndmask_b32 (vcc), dst not a source", w, 0);
        run<3>("v_cmp_lt_f64 vcc + v_cndmask_b32", w, 0); run<11>("v_cmp_lt_f64 vcc", w, 0);
        run<12>("v_cmp_lt_f64 sgpr pair", w, 0);
        run<4>("v_mov_b32", w, 0); run<5>("v_and_b32", w, 0); run<6>("v_bfi_b32", w, 0); run<10>("v_add_u32", w, 0);
        run<7>("v_add_f64 + v_cndmask_b32 (independent)", w, 0); run<8>("v_cndmask_b32 + s_nop 0", w, 0);
    }
    return 0;
}
