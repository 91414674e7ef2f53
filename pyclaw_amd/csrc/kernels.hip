// kernels.hip -- the sweep kernels + their launch code, compiled ONCE PER ARITHMETIC MODE:
//   -DPCL_NS=exact -DPCL_FAST=0 -ffp-contract=off   bit-identical to the reference (no FMA,
//                                                   correctly rounded divide/sqrt)
//   -DPCL_NS=fast  -DPCL_FAST=1 -ffp-contract=fast  FMA contraction + reciprocal-multiply
//                                                   division; checked at rtol 1e-12
// Distinct namespaces keep the two sets of template instantiations apart at link time.
#include <hip/hip_runtime.h>
#include <string>

#include "../../include/pyclaw_amd.h"
#include "classic.hpp"

namespace pcl {
namespace PCL_NS {

namespace {

int hip_fail(std::string &err, const char *what, hipError_t e) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return PCL_EHIP;
}

template <class RP, bool DIM1> int launch_x(const SweepLaunch &l, std::string &err) {
    const SweepArgs &a = l.a;
    const int nstrips = (a.mx + STRIP - 1) / STRIP;
    const int wpb = 4;
    if (a.J > 65535) { err = "more than 65535 rows"; return PCL_EINVAL; }
    const dim3 grid((unsigned)((nstrips + wpb - 1) / wpb), (unsigned)a.J);
    if (a.mcapa > 0)
        hipLaunchKernelGGL((sweep_x_kernel<RP, true, false, DIM1>), grid, dim3(256), 0, l.stream, a, nstrips);
    else
        hipLaunchKernelGGL((sweep_x_kernel<RP, false, false, DIM1>), grid, dim3(256), 0, l.stream, a, nstrips);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sweep_x launch", e);
}

template <class RP> int launch_y(const SweepLaunch &l, std::string &err) {
    const SweepArgs &a = l.a;
    const int ntiles_i = (a.I + YT_COLS - 1) / YT_COLS;
    const int ntiles_j = (a.my + STRIP - 1) / STRIP;
    const unsigned grid = (unsigned)ntiles_i * (unsigned)ntiles_j;
    if (a.mcapa > 0)
        hipLaunchKernelGGL((sweep_y_kernel<RP, true, false>), dim3(grid), dim3(256), 0, l.stream, a, ntiles_i);
    else
        hipLaunchKernelGGL((sweep_y_kernel<RP, false, false>), dim3(grid), dim3(256), 0, l.stream, a, ntiles_i);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sweep_y launch", e);
}

}  // namespace

// one directional sweep qin -> qout; ids 1 = x (or the 1-D step), 2 = y
int launch_sweep(const SweepLaunch &l, std::string &err) {
    if (l.fwave) { err = "fwave: no f-wave Riemann solver is built in yet"; return PCL_EINVAL; }
    const int rp = l.rp;
    if (l.ndim == 1) {
        if (rp == PCL_RP_ADVECTION_1D) return launch_x<Advection1D, true>(l, err);
        if (rp == PCL_RP_ACOUSTICS_1D) return launch_x<Acoustics1D, true>(l, err);
        err = "Riemann solver id is not a 1-D solver";
        return PCL_EINVAL;
    }
    if (l.ids == 1) {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch_x<Acoustics2D, false>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch_x<Euler5, false>(l, err);
    } else {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch_y<Acoustics2D>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch_y<Euler5>(l, err);
    }
    err = "Riemann solver id is not a 2-D solver";
    return PCL_EINVAL;
}

#if !PCL_FAST
__global__ void shift_test_kernel(const double *in, double *l, double *r) {
    const int t = threadIdx.x;
    const double x = in[t];
    l[t] = from_left(x);
    r[t] = from_right(x);
}
void launch_shift_test(const double *in, double *l, double *r) {
    hipLaunchKernelGGL(shift_test_kernel, dim3(1), dim3(64), 0, 0, in, l, r);
}
#endif

}  // namespace PCL_NS
}  // namespace pcl
