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
    hipStream_t stream;
};

// defined in kernels.hip, once per arithmetic mode; returns 0 or a PCL_E* code + message
namespace exact {
int launch_sweep(const SweepLaunch &l, std::string &err);
// x-pass tiles [tb_lo,tb_hi) x [ta_lo,ta_hi) that read no ghost cell; false if there are none
// ntiles[0], ntiles[1] = row tiles / tiles along a row of the x pass
bool x_interior_box(const SweepArgs &a, int box[4], int ntiles[2]);
int launch_sweep3(const SweepLaunch &l, std::string &err);   // 3-D dim-split sweep, l.ids = direction 1..3
int launch_unsplit3(const Unsplit3Launch &l, std::string &err);   // unsplit 3-D: slices + combine of one direction
int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err);  // scratch-free unsplit phase
int launch_sharp(const SweepLaunch &l, std::string &err);     // SharpClaw dq of one direction
int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err);
}
namespace fast {
int launch_sweep(const SweepLaunch &l, std::string &err);
int launch_sweep3(const SweepLaunch &l, std::string &err);
int launch_unsplit3(const Unsplit3Launch &l, std::string &err);
int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err);  // scratch-free unsplit phase
int launch_sharp(const SweepLaunch &l, std::string &err);
int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err);
}

}  // namespace pcl
