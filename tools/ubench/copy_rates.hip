// tools/ubench/copy_rates.hip -- what a read-N-planes / write-N-planes stream reaches on this chip (diagnostic, not
// part of the library): the ceiling the dimension-split sweep (80 B per cell and pass: 5 planes in, 5 planes out) is
// measured against besides the 8 TB/s data-sheet figure.
// hipcc --offload-arch=gfx950 -O3 -o copy_rates copy_rates.hip && ./copy_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// V doubles per thread and access, grid-stride; NT: nontemporal loads / stores
template <int V, bool NT> __global__ __launch_bounds__(256) void copy_k(const double *__restrict__ in, double *__restrict__ out, long n) {
    typedef double vec __attribute__((ext_vector_type(V)));
    const long nv = n / V;
    const vec *vi = (const vec *)in;
    vec *vo = (vec *)out;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
        vec v;
        if (NT) v = __builtin_nontemporal_load(vi + i); else v = vi[i];
        if (NT) __builtin_nontemporal_store(v, vo + i); else vo[i] = v;
    }
}

// the sweep's shape: one workgroup owns a tile of ROWS rows x 64 doubles of each of the 5 planes (512 B row segments,
// like the 60-cell strips + halo), loads all of it, then stores all of it
template <int ROWS, bool NT> __global__ __launch_bounds__(256) void tile_k(const double *__restrict__ in, double *__restrict__ out, int nx, int ny, long plane) {
    const int tx = blockIdx.x % (nx / 64), ty = blockIdx.x / (nx / 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double v[5][ROWS / 4];
#pragma unroll
    for (int m = 0; m < 5; m++)
#pragma unroll
        for (int r = 0; r < ROWS / 4; r++) {
            const long g = m * plane + (long)(ty * ROWS + r * 4 + w) * nx + tx * 64 + lane;
            v[m][r] = NT ? __builtin_nontemporal_load(in + g) : in[g];
        }
#pragma unroll
    for (int m = 0; m < 5; m++)
#pragma unroll
        for (int r = 0; r < ROWS / 4; r++) {
            const long g = m * plane + (long)(ty * ROWS + r * 4 + w) * nx + tx * 64 + lane;
            if (NT) __builtin_nontemporal_store(v[m][r], out + g); else out[g] = v[m][r];
        }
}

template <class F> static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 3; i++) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main() {
    const int nx = 4096, ny = 4096;
    const long plane = (long)nx * ny, n = 5 * plane;
    double *a, *b;
    CHECK(hipMalloc(&a, n * sizeof(double))); CHECK(hipMalloc(&b, n * sizeof(double)));
    CHECK(hipMemset(a, 0, n * sizeof(double))); CHECK(hipMemset(b, 0, n * sizeof(double)));
    const double bytes = 2.0 * n * sizeof(double);
    const int reps = 20;
    auto report = [&](const char *name, double ms) { printf("%-44s %8.4f ms  %7.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    report("hipMemcpyAsync D2D", time_ms([&] { CHECK(hipMemcpyAsync(b, a, n * sizeof(double), hipMemcpyDeviceToDevice, 0)); }, reps));
    for (int blocks : {2048, 8192, 32768, 131072}) {
        char nm[96];
        snprintf(nm, 96, "grid-stride double x1, %d blocks", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_k<1, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps));
        snprintf(nm, 96, "grid-stride double x2, %d blocks", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_k<2, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps));
        snprintf(nm, 96, "grid-stride double x2 nontemporal, %d blocks", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_k<2, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps));
        snprintf(nm, 96, "grid-stride double x4 nontemporal, %d blocks", blocks);
        report(nm, time_ms([&] { hipLaunchKernelGGL((copy_k<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, reps));
    }
    report("tile 16 rows x 64 x 5 planes", time_ms([&] { hipLaunchKernelGGL((tile_k<16, false>), dim3((nx / 64) * (ny / 16)), dim3(256), 0, 0, a, b, nx, ny, plane); }, reps));
    report("tile 16 rows x 64 x 5 planes nontemporal", time_ms([&] { hipLaunchKernelGGL((tile_k<16, true>), dim3((nx / 64) * (ny / 16)), dim3(256), 0, 0, a, b, nx, ny, plane); }, reps));
    report("tile 32 rows x 64 x 5 planes", time_ms([&] { hipLaunchKernelGGL((tile_k<32, false>), dim3((nx / 64) * (ny / 32)), dim3(256), 0, 0, a, b, nx, ny, plane); }, reps));
    report("tile 32 rows x 64 x 5 planes nontemporal", time_ms([&] { hipLaunchKernelGGL((tile_k<32, true>), dim3((nx / 64) * (ny / 32)), dim3(256), 0, 0, a, b, nx, ny, plane); }, reps));
    // in-place variant (the sweep writes a second array, but the y pass of a dim-split step may alias): read + write of one array
    report("tile 16 rows in place", time_ms([&] { hipLaunchKernelGGL((tile_k<16, false>), dim3((nx / 64) * (ny / 16)), dim3(256), 0, 0, a, a, nx, ny, plane); }, reps));
    CHECK(hipFree(a)); CHECK(hipFree(b));
    return 0;
}
