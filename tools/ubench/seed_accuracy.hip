// tools/ubench/seed_accuracy.hip -- relative accuracy of v_rcp_f64 / v_rsq_f64 and of the Goldschmidt by-product
// h ~ 1/(2 sqrt x) inside dsqrt (rp.hpp), measured against long double on the host.  Diagnostic only.
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o seed_accuracy seed_accuracy.hip && ./seed_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
__global__ void k(const double *x, double *o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double y = __builtin_amdgcn_rsq(v);
    double g = v * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, v);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, v);
    g = __builtin_fma(d, h, g);
    o[5 * i + 0] = y;                       // rsq seed
    o[5 * i + 1] = h;                       // h after the coupled step
    o[5 * i + 2] = g;                       // sqrt
    o[5 * i + 3] = __builtin_amdgcn_rcp(v); // rcp seed
    // 1/g from 2h with ONE Newton step against g
    double y0 = h + h;
    double e = __builtin_fma(-g, y0, 1.0);
    o[5 * i + 4] = __builtin_fma(y0, e, y0);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> x(n), o(5 * n);
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; i++) x[i] = std::exp(8.0 * (u(rng) - 0.5)) * (0.5 + u(rng));
    double *dx, *dout;
    (void)hipMalloc(&dx, n * 8); (void)hipMalloc(&dout, 5 * n * 8);
    (void)hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    (void)hipMemcpy(o.data(), dout, 5 * n * 8, hipMemcpyDeviceToHost);
    long double m[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; i++) {
        const long double v = x[i], sq = sqrtl(v);
        const long double ref[5] = {1 / sq, 0.5L / sq, sq, 1 / v, 1 / (long double)o[5 * i + 2]};
        for (int j = 0; j < 5; j++) {
            const long double e = fabsl(((long double)o[5 * i + j] - ref[j]) / ref[j]);
            if (e > m[j]) m[j] = e;
        }
    }
    const char *nm[5] = {"v_rsq_f64 seed", "h after coupled step (0.5/sqrt x)", "dsqrt result", "v_rcp_f64 seed",
                         "1/sqrt from 2h + one Newton step vs g"};
    for (int j = 0; j < 5; j++) printf("%-40s max rel err %.3Le = 2^%.1Lf\n", nm[j], m[j], log2l(m[j]));
    return 0;
}
