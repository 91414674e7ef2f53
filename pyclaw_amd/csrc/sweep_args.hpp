// sweep_args.hpp -- plain launch descriptor shared by the host TU (pclaw.hip) and the two
// kernel TUs (kernels.hip compiled once per arithmetic mode).
#pragma once
#include <hip/hip_runtime.h>
#include <string>

namespace pcl {

constexpr int MAX_WAVES_K = 8;  // == PCL_MAX_WAVES of the C ABI

struct RpParams {
    double v[8];
};

// Source term of the 2-D Euler equations with radial symmetry (test/euler/2d/shockbubble.py:59-94, the app's
// step_Euler_radial): two-stage Runge-Kutta on one cell, in the operation order of the numpy callback.  Shared by the
// stand-alone source kernel (pclaw.hip) and the y pass that applies it while storing its results (classic.hpp).
__host__ __device__ inline void euler_radial_source(double &q0, double &q1, double &q2, double &q3, double rad, double dt,
                                                    double gamma1, double ndm1) {
    const double dt2 = dt / 2.0;
    double rho = q0;
    double u = q1 / rho;
    double v = q2 / rho;
    double press = gamma1 * (q3 - 0.5 * rho * (u * u + v * v));
    const double k2 = dt2 * ndm1 / rad;
    const double s0 = q0 - k2 * q2;
    const double s1 = q1 - k2 * rho * u * v;
    const double s2 = q2 - k2 * rho * v * v;
    const double s3 = q3 - k2 * v * (q3 + press);
    rho = s0;
    u = s1 / rho;
    v = s2 / rho;
    press = gamma1 * (s3 - 0.5 * rho * (u * u + v * v));
    const double k1 = dt * ndm1 / rad;
    const double n0 = q0 - k1 * s2;
    const double n1 = q1 - k1 * rho * u * v;
    const double n2 = q2 - k1 * rho * v * v;
    const double n3 = q3 - k1 * v * (s3 + press);
    q0 = n0; q1 = n1; q2 = n2; q3 = n3;
}

// SharpClaw form of the same source (apps/euler/2d/shockbubble/shockbubble.py:95-122, dq_Euler_radial): the increment
// dt * psi(q) of one cell, in the operation order of the numpy callback (-dt*(ndim-1)/rad is one array, then the
// products left to right); the energy-equation tracer component gets 0.
__host__ __device__ inline void euler_radial_dq(double q0, double q1, double q2, double q3, double rad, double dt,
                                                double gamma1, double ndm1, double (&d)[4]) {
    const double rho = q0;
    const double u = q1 / rho;
    const double v = q2 / rho;
    const double press = gamma1 * (q3 - 0.5 * rho * (u * u + v * v));
    const double c = -dt * ndm1 / rad;
    d[0] = c * q2;
    d[1] = c * rho * u * v;
    d[2] = c * rho * v * v;
    d[3] = c * v * (q3 + press);
}

struct SweepArgs {
    const double *qin;
    double *qout;
    const double *aux;
    long pitch;       // doubles between rows
    long plane;       // doubles between components
    int I, J;         // cells per row / rows, ghost cells included
    int mbc, mx, my;  // interior extents
    int mcapa;        // 0 = none, else 1-based aux component (method(6))
    int order;        // method(2)
    int mthlim[MAX_WAVES_K];
    double dtd;       // dt/dx of the sweep direction
    double dt, dx;    // separately, for the 1-D capa form dt/(dx*capa) (step1.f:70)
    RpParams par;
    unsigned long long *cfl;  // device word holding the running max (as ordered bits)
    int xcd;          // XCD-aware block order in the kernels whose tiles share cache lines (classic.hpp)
    int ablate;       // diagnostic only (tools/kbench.py): bit0 = skip the arithmetic (copy through)
    // "virtual ghost cells": the x pass of the dim-split step evaluates the boundary conditions while it
    // loads (a ghost cell is an index remap of an interior cell, or a constant): no ghost-fill launches.
    // vbc[2*idim+side] = -1 (read memory) | PCL_BC_OUTFLOW | PERIODIC | REFLECTING | PCL_BC_CUSTOM (= vconst)
    int vbc_on;
    int vbc[4];
    double vconst[4][8];
    // Tile subset of the x pass, for overlapping the halo exchange with the interior (pclaw.hip,
    // step_split_overlapped): 0 = every tile, 1 = only tiles inside box = [tb_lo,tb_hi) x [ta_lo,ta_hi)
    // (they read no ghost cell), 2 = only the tiles outside the box.
    int sub;
    int box[4];
    // 3-D dimension-split sweeps (sweep3_kernel) only: strides in doubles along the sweep / across a tile /
    // per batch (blockIdx.y), extents with ghost cells, interior cells along the sweep, and the inclusive
    // ranges of across / batch indices whose slices are swept (the others are copied through)
    long s_al, s_ac, s_b;
    int n_al, n_ac, n_b, m_al;
    int lo_ac, hi_ac, lo_b, hi_b;
    // SharpClaw: Runge-Kutta combination fused into the LAST directional pass (sharp_kernel store phase):
    // rk_d = op(rk_a, rk_b, dq) per interior cell instead of writing dq (ops 1, 2, 5 of rk_kernel)
    int rk_op;
    const double *rk_a, *rk_b;
    double *rk_d;
    double rk_ca, rk_cb, rk_cc;
    // unsplit algorithm (step2.f) only:
    int trans;        // method(3): 0 no transverse terms, 1 increment waves, 2 + correction waves
    double dtd_t;     // dt/d of the transverse direction
    // (kept at the END of the block: the kernels' scalar loads of the fields above keep their offsets)
    // source term applied by the LAST pass of a dimension-split step while it stores its results (Godunov splitting,
    // clawpack.py:156-159): 0 none, 1 euler_radial_source(gamma1, ndim-1), aux plane 0 = radial coordinate.  SharpClaw:
    // 1 = euler_radial_dq added to deltaq by the last pass of a stage (sharpclaw.py:232-235, dq_src)
    int src_id;
    double src_p[2];
};

// unsplit 3-D (classic3.hpp): one direction's slices into 14 scratch plane sets, then the ordered combine
struct Unsplit3Launch {
    SweepArgs a;          // qin = qold; s_al/n_al/m_al, dtd, par, mthlim, order, cfl as for sweep3
    double *scr[14];
    double *qacc;         // the new state being accumulated
    long s_e, s_f;
    int n_e, n_f, m_e, m_f;
    int m3, m4;
    double dty, dtz;
    int dir;              // 1..3
    int rp;
    hipStream_t stream;
};

struct RkLaunch {
    double *d;
    const double *a, *b, *c;
    double ca, cb, cc;
    long n;
    int op;
};

struct SweepLaunch {
    SweepArgs a;
    int ndim;   // 1 or 2
    int rp;     // PCL_RP_*
    int ids;    // 1 = x pass (or the 1-D step), 2 = y pass
    int fwave;
    int lim_type;  // SharpClaw reconstruction (2 PyWENO weno5, 3 legacy weno5)
    int char_decomp = 0;   // SharpClaw: 1 = wave-based reconstruction (1-D: tvd2_wave / weno5_wave)
    hipStream_t stream;
};

// defined in kernels.hip, once per arithmetic mode; returns 0 or a PCL_E* code + message
namespace exact {
int launch_sweep(const SweepLaunch &l, std::string &err);
int launch_step2ds(const SweepLaunch &l, std::string &err);   // whole dim-split 2-D step in one kernel (classic_fused.hpp)
// x-pass tiles [tb_lo,tb_hi) x [ta_lo,ta_hi) that read no ghost cell; false if there are none
// ntiles[0], ntiles[1] = row tiles / tiles along a row of the x pass
bool x_interior_box(const SweepArgs &a, int box[4], int ntiles[2]);
bool step2ds_interior_box(const SweepArgs &a, int box[4], int ntiles[2]);   // the same for the one-kernel step: (nty, ntx)
int launch_sweep3(const SweepLaunch &l, std::string &err);   // 3-D dim-split sweep, l.ids = direction 1..3
int launch_unsplit3(const Unsplit3Launch &l, std::string &err);   // unsplit 3-D: slices + combine of one direction
int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err);  // scratch-free unsplit phase
int launch_sharp(const SweepLaunch &l, std::string &err);     // SharpClaw dq of one direction
int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err);
}
namespace fast {
int launch_sweep(const SweepLaunch &l, std::string &err);
int launch_step2ds(const SweepLaunch &l, std::string &err);   // whole dim-split 2-D step in one kernel (classic_fused.hpp)
int launch_sweep3(const SweepLaunch &l, std::string &err);
int launch_unsplit3(const Unsplit3Launch &l, std::string &err);
int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err);  // scratch-free unsplit phase
int launch_sharp(const SweepLaunch &l, std::string &err);
int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err);
}
// exact + IEEE quotients for underflow-range numerators as well (rp.hpp PCL_DENORM_GUARD), PCL_MATH_STRICT
namespace strict {
int launch_sweep(const SweepLaunch &l, std::string &err);
int launch_step2ds(const SweepLaunch &l, std::string &err);   // whole dim-split 2-D step in one kernel (classic_fused.hpp)
int launch_sweep3(const SweepLaunch &l, std::string &err);
int launch_unsplit3(const Unsplit3Launch &l, std::string &err);
int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err);
int launch_sharp(const SweepLaunch &l, std::string &err);
int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err);
}

}  // namespace pcl
