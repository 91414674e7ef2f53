// kernels.hip -- the sweep kernels + their launch code, compiled ONCE PER ARITHMETIC MODE:
//   -DPCL_NS=exact -DPCL_FAST=0 -ffp-contract=off   bit-identical to the reference (no FMA,
//                                                   correctly rounded divide/sqrt)
//   -DPCL_NS=fast  -DPCL_FAST=1 -ffp-contract=fast  FMA contraction + reciprocal-multiply
//                                                   division; checked at rtol 1e-12
// Distinct namespaces keep the two sets of template instantiations apart at link time.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include <string>

#include "../../include/pyclaw_amd.h"
#include "classic.hpp"
#include "sharpclaw.hpp"
#include "classic3.hpp"
#include "classic_fused.hpp"

namespace pcl {
namespace PCL_NS {

namespace {

int hip_fail(std::string &err, const char *what, hipError_t e) {
    err = std::string(what) + ": " + hipGetErrorString(e);
    return PCL_EHIP;
}

template <class RP, int IXY, bool DIM1> int launch(const SweepLaunch &l, std::string &err) {
    using T = TileShape<IXY>;
    const SweepArgs &a = l.a;
    const int n_across = IXY == 1 ? a.J : a.I + (LINE - a.mbc);  // y: columns counted from the line boundary
    const int m_along = IXY == 1 ? a.mx : a.my;
    const int ntiles_across = (n_across + T::ACROSS - 1) / T::ACROSS;
    const int ntiles_along = (m_along + T::NSTRIP * STRIP - 1) / (T::NSTRIP * STRIP);
    unsigned nblocks = (unsigned)ntiles_across * (unsigned)ntiles_along;
    if (a.sub != 0) {
        if (IXY != 1 || DIM1 || a.box[0] < 0 || a.box[1] > ntiles_across || a.box[2] < 0 ||
            a.box[3] > ntiles_along || a.box[0] >= a.box[1] || a.box[2] >= a.box[3]) {
            err = "tile subset: bad box";
            return PCL_EINVAL;
        }
        const unsigned inside = (unsigned)(a.box[1] - a.box[0]) * (unsigned)(a.box[3] - a.box[2]);
        nblocks = a.sub == 1 ? inside : nblocks - inside;
        if (nblocks == 0) return PCL_OK;
    }
    const dim3 grid(nblocks);
    constexpr bool FW = IsFwave<RP>::value;       // f-wave solvers run the flux2fw.f / step1fw.f form of the kernel
    if ((l.fwave != 0) != FW) {
        err = FW ? "this Riemann solver returns f-waves: set solver.fwave = True (classic1fw / classic2fw)"
                 : "solver.fwave = True needs an f-wave Riemann solver (elasticity_fwave_1d, psystem_fwave_2d)";
        return PCL_EINVAL;
    }
    if (a.src_id != 0) {      // fused source term: its own instantiation (classic.hpp), Euler y pass without capa
        if constexpr (IXY == 2 && !DIM1 && std::is_same<RP, Euler5>::value) {
            if (a.src_id != 1 || a.mcapa > 0) { err = "fused source: Euler radial source, no capacity function"; return PCL_EINVAL; }
            hipLaunchKernelGGL((sweep_kernel<RP, IXY, false, FW, DIM1, true>), grid, dim3(256), 0, l.stream, a,
                               ntiles_across, ntiles_along);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? PCL_OK : hip_fail(err, "sweep launch", e);
        } else {
            err = "fused source: only the y pass of the Euler solver has it";
            return PCL_EINVAL;
        }
    }
    if (a.mcapa > 0)
        hipLaunchKernelGGL((sweep_kernel<RP, IXY, true, FW, DIM1>), grid, dim3(256), 0, l.stream, a,
                           ntiles_across, ntiles_along);
    else
        hipLaunchKernelGGL((sweep_kernel<RP, IXY, false, FW, DIM1>), grid, dim3(256), 0, l.stream, a,
                           ntiles_across, ntiles_along);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sweep launch", e);
}

}  // namespace

// one directional sweep qin -> qout; ids 1 = x (or the 1-D step), 2 = y
int launch_sweep(const SweepLaunch &l, std::string &err) {
    const int rp = l.rp;
    if (l.ndim == 1) {
        if (rp == PCL_RP_ADVECTION_1D) return launch<Advection1D, 1, true>(l, err);
        if (rp == PCL_RP_ACOUSTICS_1D) return launch<Acoustics1D, 1, true>(l, err);
        if (rp == PCL_RP_BURGERS_1D) return launch<Burgers1D, 1, true>(l, err);
        if (rp == PCL_RP_EULER_1D) return launch<Euler1D, 1, true>(l, err);
        if (rp == PCL_RP_SHALLOW_1D) return launch<Shallow1D, 1, true>(l, err);
        if (rp == PCL_RP_ADVECTION_COLOR_1D) return launch<AdvectionColor1D, 1, true>(l, err);
        if (rp == PCL_RP_ELASTICITY_FWAVE_1D) return launch<Elasticity1D, 1, true>(l, err);
        err = "Riemann solver id is not a 1-D solver";
        return PCL_EINVAL;
    }
    if (l.ids == 1) {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch<Acoustics2D, 1, false>(l, err);
        if (rp == PCL_RP_ADVECTION_2D) return launch<Advection2D, 1, false>(l, err);
        if (rp == PCL_RP_SHALLOW_2D) return launch<Shallow2D, 1, false>(l, err);
        if (rp == PCL_RP_VC_ACOUSTICS_2D) return launch<VcAcoustics2D, 1, false>(l, err);
        if (rp == PCL_RP_VC_ADVECTION_2D) return launch<VcAdvection2D, 1, false>(l, err);
        if (rp == PCL_RP_SHALLOW_SPHERE_2D) return launch<ShallowSphere, 1, false>(l, err);
        if (rp == PCL_RP_PSYSTEM_FWAVE_2D) return launch<PSystem2D, 1, false>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch<Euler5, 1, false>(l, err);
    } else {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch<Acoustics2D, 2, false>(l, err);
        if (rp == PCL_RP_ADVECTION_2D) return launch<Advection2D, 2, false>(l, err);
        if (rp == PCL_RP_SHALLOW_2D) return launch<Shallow2D, 2, false>(l, err);
        if (rp == PCL_RP_VC_ACOUSTICS_2D) return launch<VcAcoustics2D, 2, false>(l, err);
        if (rp == PCL_RP_VC_ADVECTION_2D) return launch<VcAdvection2D, 2, false>(l, err);
        if (rp == PCL_RP_SHALLOW_SPHERE_2D) return launch<ShallowSphere, 2, false>(l, err);
        if (rp == PCL_RP_PSYSTEM_FWAVE_2D) return launch<PSystem2D, 2, false>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch<Euler5, 2, false>(l, err);
    }
    err = "Riemann solver id is not a 2-D solver";
    return PCL_EINVAL;
}

// the whole dimension-split 2-D step, qin -> qout (classic_fused.hpp); l.a as for the x pass + a.dtd_t = dt/dy and
// a.src_id of the step.  The host (pclaw.hip: fused_step_ok) only sends what the kernel covers.
namespace {
template <class RP> int launch_step2ds_t(const SweepLaunch &l, std::string &err) {
    const SweepArgs &a = l.a;
    constexpr bool FW = IsFwave<RP>::value;
    if ((l.fwave != 0) != FW) { err = "solver.fwave does not match the Riemann solver"; return PCL_EINVAL; }
    if (a.mbc != HALO || a.mcapa > 0) { err = "step2ds kernel: mbc = 2, no capacity function"; return PCL_EINVAL; }
    const int ntx = (a.mx + F_OWN_C - 1) / F_OWN_C, nty = (a.my + F_OWN_R - 1) / F_OWN_R;
    unsigned nblocks = (unsigned)ntx * (unsigned)nty;
    if (a.sub != 0) {
        if (a.box[0] < 0 || a.box[1] > nty || a.box[2] < 0 || a.box[3] > ntx || a.box[0] >= a.box[1] || a.box[2] >= a.box[3]) {
            err = "step2ds tile subset: bad box";
            return PCL_EINVAL;
        }
        const unsigned inside = (unsigned)(a.box[1] - a.box[0]) * (unsigned)(a.box[3] - a.box[2]);
        nblocks = a.sub == 1 ? inside : nblocks - inside;
        if (nblocks == 0) return PCL_OK;
    }
    const dim3 grid(nblocks);
    if (a.src_id != 0) {
        if constexpr (std::is_same<RP, Euler5>::value) {
            if (a.src_id != 1) { err = "fused source: Euler radial source"; return PCL_EINVAL; }
            hipLaunchKernelGGL((step2ds_kernel<RP, FW, true>), grid, dim3(F_THREADS), 0, l.stream, a, ntx, nty);
        } else {
            err = "fused source: only the Euler solver has it";
            return PCL_EINVAL;
        }
    } else
        hipLaunchKernelGGL((step2ds_kernel<RP, FW, false>), grid, dim3(F_THREADS), 0, l.stream, a, ntx, nty);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "step2ds launch", e);
}
}  // namespace
int launch_step2ds(const SweepLaunch &l, std::string &err) {
    if (l.rp == PCL_RP_EULER5_2D) return launch_step2ds_t<Euler5>(l, err);
    if (l.rp == PCL_RP_ACOUSTICS_2D) return launch_step2ds_t<Acoustics2D>(l, err);
    if (l.rp == PCL_RP_ADVECTION_2D) return launch_step2ds_t<Advection2D>(l, err);
    if (l.rp == PCL_RP_SHALLOW_2D) return launch_step2ds_t<Shallow2D>(l, err);
    err = "step2ds kernel: Riemann solvers without aux arrays only";
    return PCL_EINVAL;
}

// 3-D dimension-split sweep along direction l.ids (classic.hpp: sweep3_kernel)
int launch_sweep3(const SweepLaunch &l, std::string &err) {
    const SweepArgs &a = l.a;
    if (l.fwave) { err = "fwave: no 3-D f-wave Riemann solver is built in"; return PCL_EINVAL; }
    if (l.rp != PCL_RP_VC_ACOUSTICS_3D) { err = "Riemann solver id is not a 3-D solver"; return PCL_EINVAL; }
    const int n_ac = l.ids == 1 ? a.n_ac : a.n_ac + (LINE - a.mbc);  // y, z: columns counted from the line boundary
    const int ac = l.ids == 1 ? 4 : 16, adv = l.ids == 1 ? 4 * STRIP : STRIP;   // tile shape, classic.hpp
    const int ntiles_ac = (n_ac + ac - 1) / ac, ntiles_al = (a.m_al + adv - 1) / adv;
    const dim3 grid((unsigned)ntiles_ac * (unsigned)ntiles_al, (unsigned)a.n_b);
    if (a.mcapa > 0) {      // capacity function (step3ds.f:138-141,196-200)
        if (l.ids == 1) hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 1, true>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
        else if (l.ids == 2) hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 2, true>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
        else hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 3, true>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
    } else if (l.ids == 1) hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 1>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
    else if (l.ids == 2) hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 2>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
    else hipLaunchKernelGGL((sweep3_kernel<VcAcoustics3D, 3>), grid, dim3(256), 0, l.stream, a, ntiles_ac, ntiles_al);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sweep3 launch", e);
}

// unsplit 3-D step, one direction: slices3_kernel into the scratch planes, then combine3_kernel (classic3.hpp)
namespace {
template <class RP, int DIR> int launch_unsplit3_t(const Unsplit3Launch &l, std::string &err) {
    const SweepArgs &a = l.a;
    Slices3Args t;
    for (int k = 0; k < 14; k++) t.scr[k] = l.scr[k];
    t.s_e = l.s_e; t.s_f = l.s_f; t.n_e = l.n_e; t.n_f = l.n_f;
    t.lo_e = a.mbc - 1; t.hi_e = a.mbc + l.m_e; t.lo_f = a.mbc - 1; t.hi_f = a.mbc + l.m_f;
    t.m3 = l.m3; t.m4 = l.m4; t.dty = l.dty; t.dtz = l.dtz;
    const int ntiles_al = (a.m_al + STRIP - 1) / STRIP;
    static const int marching = [] { const char *e = getenv("PCL_TUNE_UNSPLIT3"); return e ? atoi(e) : 1; }();
    if (marching) {
        // the marching form (classic3.hpp: march3_kernel): no scratch planes, one launch per direction
        constexpr int NW = 8;
        March3Args g;
        g.qsrc = DIR == 1 ? a.qin : l.qacc;
        g.qacc = l.qacc;
        const bool e_outer = DIR == 2;
        g.s_w = e_outer ? l.s_f : l.s_e; g.s_m = e_outer ? l.s_e : l.s_f;
        g.n_w = e_outer ? l.n_f : l.n_e; g.n_m = e_outer ? l.n_e : l.n_f;
        g.m_w = e_outer ? l.m_f : l.m_e; g.m_m = e_outer ? l.m_e : l.m_f;
        g.ntiles_al = ntiles_al;
        g.ntiles_w = (g.m_w + (NW - 2) - 1) / (NW - 2);
        // march segments: enough workgroups to fill the 256 CUs in whole rounds, not so many that the two extra source
        // planes per segment cost more than the idle CUs would
        const long per = (long)g.ntiles_al * g.ntiles_w;
        int best = 1;
        double best_eff = 0.0;
        for (int ns = 1; ns <= (g.m_m + 7) / 8; ns++) {
            const int seg = (g.m_m + ns - 1) / ns;
            const long wgs = per * ((g.m_m + seg - 1) / seg);
            const double eff = (double)wgs / (256.0 * ((wgs + 255) / 256)) * seg / (seg + 2.0);
            if (eff > best_eff + 1e-9) { best_eff = eff; best = ns; }
        }
        g.seg = (g.m_m + best - 1) / best;
        const int nseg = (g.m_m + g.seg - 1) / g.seg;
        if (DIR == 1) {     // the first direction writes interior cells only: the ghost frame of the result := qold's
            const int I = a.n_al, J = l.n_e, K = l.n_f;
            hipLaunchKernelGGL(ghost3_copy_kernel, dim3((unsigned)((I + 255) / 256), (unsigned)J, (unsigned)K), dim3(256), 0,
                               l.stream, a.qin, l.qacc, (int)RP::MEQN, a.plane, I, J, K, l.s_e, a.mbc);
        }
        static const int dense = [] { const char *e = getenv("PCL_TUNE_UNSPLIT3_DENSE"); return e ? atoi(e) : 0; }();
        if (RP::T3_PRESSURE && !dense && l.m3 == 2 && l.m4 == 2)      // the solver default, method(3) = 22, as constants
            hipLaunchKernelGGL((march3p_kernel<RP, DIR, NW, 22>), dim3((unsigned)(per * nseg)), dim3(NW * WAVE), 0, l.stream, a, t, g);
        else if (RP::T3_PRESSURE && !dense)
            hipLaunchKernelGGL((march3p_kernel<RP, DIR, NW>), dim3((unsigned)(per * nseg)), dim3(NW * WAVE), 0, l.stream, a, t, g);
        else
            hipLaunchKernelGGL((march3_kernel<RP, DIR, NW>), dim3((unsigned)(per * nseg)), dim3(NW * WAVE), 0, l.stream, a, t, g);
        hipError_t e = hipGetLastError();
        return e == hipSuccess ? PCL_OK : hip_fail(err, "march3 launch", e);
    }
    // workgroups of 4 strips: 4 consecutive y-like rows (x direction) / 4 consecutive i (y, z directions; classic3.hpp)
    const int n4 = DIR == 2 ? l.m_f + 2 : l.m_e + 2, nother = DIR == 2 ? l.m_e + 2 : l.m_f + 2;
    hipLaunchKernelGGL((slices3_kernel<RP, DIR>), dim3((unsigned)ntiles_al * ((n4 + 3) / 4), (unsigned)nother), dim3(256), 0,
                       l.stream, a, t, ntiles_al);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(err, "slices3 launch", e);
    Combine3Args c;
    for (int k = 0; k < 14; k++) c.scr[k] = l.scr[k];
    c.qsrc = a.qin; c.qacc = l.qacc; c.plane = a.plane; c.s_al = a.s_al; c.s_e = l.s_e; c.s_f = l.s_f;
    c.n_al = a.n_al; c.n_e = l.n_e; c.n_f = l.n_f; c.mbc = a.mbc; c.m_al = a.m_al; c.m_e = l.m_e; c.m_f = l.m_f;
    c.meqn = RP::MEQN; c.dtd = a.dtd; c.dty = l.dty; c.dtz = l.dtz;
    c.e_outer = DIR == 2 ? 1 : 0;       // step3.f: y sweeps loop k (their y-like index) outside, x and z sweeps inside
    c.first = DIR == 1 ? 1 : 0;
    c.dir = DIR;
    const int ni = DIR == 1 ? a.n_al : (DIR == 2 ? l.n_f : l.n_e);     // physical extents (i, j, k) of the block
    const int nj = DIR == 1 ? l.n_e : (DIR == 2 ? a.n_al : l.n_f);
    const int nk = DIR == 1 ? l.n_f : (DIR == 2 ? l.n_e : a.n_al);
    hipLaunchKernelGGL(combine3_kernel, dim3((unsigned)((ni + 255) / 256), (unsigned)nj, (unsigned)nk), dim3(256), 0,
                       l.stream, c);
    e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "combine3 launch", e);
}
}  // namespace

int launch_unsplit3(const Unsplit3Launch &l, std::string &err) {
    if (l.rp != PCL_RP_VC_ACOUSTICS_3D) { err = "Riemann solver id is not a 3-D solver with transverse solvers"; return PCL_EINVAL; }
    if (l.a.mcapa > 0) { err = "3-D: capacity function not implemented"; return PCL_EINVAL; }
    if (l.dir == 1) return launch_unsplit3_t<VcAcoustics3D, 1>(l, err);
    if (l.dir == 2) return launch_unsplit3_t<VcAcoustics3D, 2>(l, err);
    return launch_unsplit3_t<VcAcoustics3D, 3>(l, err);
}

#if !PCL_FAST
// the same for the one-kernel step (classic_fused.hpp): tile (ty, tx) reads rows mbc-2+F_OWN_R*ty .. +F_ROWS-1 and columns
// mbc-2+60*tx .. +63; box = [ty_lo, ty_hi) x [tx_lo, tx_hi), ntiles = (nty, ntx)
bool step2ds_interior_box(const SweepArgs &a, int box[4], int ntiles[2]) {
    const int ntx = (a.mx + F_OWN_C - 1) / F_OWN_C, nty = (a.my + F_OWN_R - 1) / F_OWN_R;
    box[0] = 1;
    const int roomy = a.my + HALO - F_ROWS;       // F_OWN_R*ty + F_ROWS <= my + 2
    box[1] = roomy < 0 ? 0 : std::min(nty, roomy / F_OWN_R + 1);
    box[2] = 1;
    const int roomx = a.mx + HALO - F_COLS;
    box[3] = roomx < 0 ? 0 : std::min(ntx, roomx / F_OWN_C + 1);
    ntiles[0] = nty;
    ntiles[1] = ntx;
    return box[0] < box[1] && box[2] < box[3];
}
bool x_interior_box(const SweepArgs &a, int box[4], int ntiles[2]) {
    using T = TileShape<1>;
    constexpr int ADV = T::NSTRIP * STRIP;  // cells a tile advances along the row
    const int ntb = (a.J + T::ACROSS - 1) / T::ACROSS, nta = (a.mx + ADV - 1) / ADV;
    // rows b0 = ACROSS*tb .. b0+ACROSS-1 inside [mbc, J-mbc); cells a0 = mbc-HALO+ADV*ta .. a0+ALONG-1 inside [mbc, I-mbc)
    box[0] = (a.mbc + T::ACROSS - 1) / T::ACROSS;
    box[1] = std::min(ntb, (a.J - a.mbc) / T::ACROSS);
    box[2] = (HALO + ADV - 1) / ADV;
    const int room = a.I - a.mbc - T::ALONG - (a.mbc - HALO);  // ADV*ta <= room
    box[3] = room < 0 ? 0 : std::min(nta, room / ADV + 1);
    ntiles[0] = ntb;
    ntiles[1] = nta;
    return box[0] < box[1] && box[2] < box[3];
}
#endif

// unsplit step (step2.f / step2qcor.f): both phases without scratch planes (classic.hpp: unsplit_x/y_kernel)
namespace {
template <class RP, int UX, int UY> int launch_unsplit_u(const SweepLaunch &l, const double *qx, std::string &err) {
    SweepArgs a = l.a;
    if (l.ids == 1) {
        const int nstrips = (a.mx + STRIP - 1) / STRIP;
        const int ntr = (a.my + (UX - 2) - 1) / (UX - 2);
        if (a.sub != 0) {
            // tiles whose 64 cells x UX rows lie inside the interior: cells mbc-2+60*ta .. +63 within [mbc, mbc+mx),
            // rows mbc-1+(UX-2)*tr .. +UX-1 within [mbc, mbc+my)
            a.box[0] = 1;
            a.box[1] = a.my - UX + 1 >= 0 ? (a.my - UX + 1) / (UX - 2) + 1 : 0;
            a.box[2] = 1;
            a.box[3] = a.mx >= 62 ? (a.mx - 62) / STRIP + 1 : 0;
        }
        if (a.mcapa > 0)
            hipLaunchKernelGGL((unsplit_x_kernel<RP, IsFwave<RP>::value, UX, true>), dim3((unsigned)nstrips * ntr),
                               dim3(UX * WAVE), 0, l.stream, a, nstrips);
        else
            hipLaunchKernelGGL((unsplit_x_kernel<RP, IsFwave<RP>::value, UX>), dim3((unsigned)nstrips * ntr),
                               dim3(UX * WAVE), 0, l.stream, a, nstrips);
    } else {
        const int nti = (a.mx + (UY - 2) - 1) / (UY - 2);
        const int ntj = (a.my + STRIP - 1) / STRIP;
        static const int ymarch = [] { const char *e = getenv("PCL_TUNE_YMARCH"); return e ? atoi(e) : 1; }();
        if constexpr (RP::NAUX == 0 && UY == 16) {
            // marching y phase (classic.hpp: unsplit_ym_kernel): segments of `seg` lines of 16 columns; enough
            // workgroups for ~4 rounds over the 256 CUs, warm-up step of a segment <= 1/8 of its work
            const int nlines = (a.mx + 15) / 16;
            int nseg = (1024 + ntj - 1) / ntj;
            if (nseg > (nlines + 7) / 8) nseg = (nlines + 7) / 8;
            if (nseg < 1) nseg = 1;
            const int seg = (nlines + nseg - 1) / nseg;
            nseg = (nlines + seg - 1) / seg;
            // a small grid cannot give every CU two marching workgroups (1024^2: 18 bands x 8 segments = 144; 2048^2: 560): the tile
            // kernel's many independent workgroups are the better shape there (C2, acoustics 1024^2: 103 vs 131 us per step; 2048^2: 240 vs 249; 4096^2: 731 vs 702, same box)
            const bool big = (long)ntj * nseg >= 1024 || ymarch == 8 || ymarch == 16;
            if (ymarch && big && a.mcapa <= 0 && (a.src_id == 0 || std::is_same<RP, Euler5>::value)) {
                // wavefronts per workgroup: 8 (two slices each per step, 256 VGPRs: no spills) for the register-heavy
                // Euler core, 16 for the small systems; PCL_TUNE_YMARCH=16 / 8 forces one
                constexpr bool heavy = RP::MEQN >= 5;
                const bool eight = ymarch == 8 || (ymarch != 16 && heavy);
                if (a.src_id != 0) {
                    if constexpr (std::is_same<RP, Euler5>::value) {
                        if (eight) hipLaunchKernelGGL((unsplit_ym_kernel<RP, false, true, 8>), dim3((unsigned)ntj * nseg), dim3(8 * WAVE), 0, l.stream, a, ntj, seg, qx);
                        else hipLaunchKernelGGL((unsplit_ym_kernel<RP, false, true, 16>), dim3((unsigned)ntj * nseg), dim3(16 * WAVE), 0, l.stream, a, ntj, seg, qx);
                    }
                } else if (eight)
                    hipLaunchKernelGGL((unsplit_ym_kernel<RP, IsFwave<RP>::value, false, 8>), dim3((unsigned)ntj * nseg),
                                       dim3(8 * WAVE), 0, l.stream, a, ntj, seg, qx);
                else
                    hipLaunchKernelGGL((unsplit_ym_kernel<RP, IsFwave<RP>::value, false, 16>), dim3((unsigned)ntj * nseg),
                                       dim3(16 * WAVE), 0, l.stream, a, ntj, seg, qx);
                hipError_t e = hipGetLastError();
                if (e != hipSuccess) return hip_fail(err, "unsplit y (marching) launch", e);
                return PCL_OK;
            }
        }
        if (a.src_id != 0) {       // fused source term: Euler solver without a capacity function
            if constexpr (std::is_same<RP, Euler5>::value) {
                if (a.src_id != 1 || a.mcapa > 0) { err = "fused source: Euler radial source, no capacity function"; return PCL_EINVAL; }
                hipLaunchKernelGGL((unsplit_y_kernel<RP, false, UY, false, true>), dim3((unsigned)nti * ntj),
                                   dim3(UY * WAVE), 0, l.stream, a, nti, qx);
            } else {
                err = "fused source: only the Euler solver has it";
                return PCL_EINVAL;
            }
        } else if (a.mcapa > 0)
            hipLaunchKernelGGL((unsplit_y_kernel<RP, IsFwave<RP>::value, UY, true>), dim3((unsigned)nti * ntj),
                               dim3(UY * WAVE), 0, l.stream, a, nti, qx);
        else
            hipLaunchKernelGGL((unsplit_y_kernel<RP, IsFwave<RP>::value, UY>), dim3((unsigned)nti * ntj),
                               dim3(UY * WAVE), 0, l.stream, a, nti, qx);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "unsplit launch", e);
}
template <class RP> int launch_unsplit_t(const SweepLaunch &l, const double *qx, std::string &err) {
    static const int var = [] { const char *e = getenv("PCL_TUNE_UNSPLIT"); return e ? atoi(e) : 0; }();
    switch (var) {   // slices per workgroup in the x / y phase (tools/kbench.py A/B)
    case 1: return launch_unsplit_u<RP, 8, 8>(l, qx, err);
    case 2: return launch_unsplit_u<RP, 16, 8>(l, qx, err);
    case 3: return launch_unsplit_u<RP, 8, 16>(l, qx, err);
    case 4: return launch_unsplit_u<RP, 12, 12>(l, qx, err);
    case 5: return launch_unsplit_u<RP, 16, 16>(l, qx, err);
    default:
        // 16 wavefronts per workgroup leave 128 VGPRs per lane.  The Euler kernels spill ~20 registers there and are
        // still 4 % faster than with 12 wavefronts (170 VGPRs, no spills) -- except with a capacity function, where the
        // spill stores reach HBM (2.4 GB written per x launch at 4096^2 for 0.67 GB of results,
        // profiles/r02_pmc_hbm.json) at equal speed: 12 there.  The sphere solver (9 + 27 aux values per lane) takes 8.
        if constexpr (RP::NAUX >= 9) return launch_unsplit_u<RP, 8, 8>(l, qx, err);
        else if constexpr (RP::MEQN >= 5) {
            if (l.a.mcapa > 0) return launch_unsplit_u<RP, 12, 12>(l, qx, err);
            return launch_unsplit_u<RP, 16, 16>(l, qx, err);
        } else return launch_unsplit_u<RP, 16, 16>(l, qx, err);
    }
}
}  // namespace

int launch_unsplit(const SweepLaunch &l, const double *qx, std::string &err) {
    if ((l.fwave != 0) != (l.rp == PCL_RP_PSYSTEM_FWAVE_2D)) {
        err = "solver.fwave must be True for an f-wave Riemann solver and only for one";
        return PCL_EINVAL;
    }
    if (l.rp == PCL_RP_PSYSTEM_FWAVE_2D) return launch_unsplit_t<PSystem2D>(l, qx, err);
    if (l.rp == PCL_RP_ACOUSTICS_2D) return launch_unsplit_t<Acoustics2D>(l, qx, err);
    if (l.rp == PCL_RP_ADVECTION_2D) return launch_unsplit_t<Advection2D>(l, qx, err);
    if (l.rp == PCL_RP_SHALLOW_2D) return launch_unsplit_t<Shallow2D>(l, qx, err);
    if (l.rp == PCL_RP_VC_ACOUSTICS_2D) return launch_unsplit_t<VcAcoustics2D>(l, qx, err);
    if (l.rp == PCL_RP_VC_ADVECTION_2D) return launch_unsplit_t<VcAdvection2D>(l, qx, err);
    if (l.rp == PCL_RP_EULER5_2D) return launch_unsplit_t<Euler5>(l, qx, err);
    if (l.rp == PCL_RP_SHALLOW_SPHERE_2D) return launch_unsplit_t<ShallowSphere>(l, qx, err);
    err = "Riemann solver id is not a 2-D solver";
    return PCL_EINVAL;
}

// SharpClaw: dq of one direction (x writes dq, y accumulates)
namespace {
template <class RP, int IXY, int K> int launch_sharp_k(const SweepLaunch &l, std::string &err) {
    SweepArgs a = l.a;
    constexpr int SH = K, SS = sstrip(K);
    if (IXY == 1 && a.sub != 0) {
        // x-pass tiles whose 16 rows x 64 cells lie inside the interior: rows 16*tb .. +15 within [mbc, mbc+my), cells
        // mbc-K+SS*ta .. +63 within [mbc, mbc+mx)
        a.box[0] = (a.mbc + T_ACROSS_S - 1) / T_ACROSS_S;
        a.box[1] = a.mbc + a.my >= T_ACROSS_S ? (a.mbc + a.my - T_ACROSS_S) / T_ACROSS_S + 1 : 0;
        a.box[2] = 1;
        a.box[3] = a.mx >= WAVE - SH ? (a.mx - (WAVE - SH)) / SS + 1 : 0;
    } else a.sub = 0;
    const int n_across = IXY == 1 ? a.J : a.I + (LINE - a.mbc);
    const int m_along = IXY == 1 ? a.mx : a.my;
    constexpr int TA = sharp_tile_across<RP, IXY>();
    const int ntiles_across = (n_across + TA - 1) / TA;
    const int ntiles_along = (m_along + SS - 1) / SS;
    const dim3 grid((unsigned)ntiles_across * (unsigned)ntiles_along);
    const bool capa = a.mcapa > 0;
#define PCL_SHARP_LAUNCH(CAPA_, LIM_)                                                                  \
    hipLaunchKernelGGL((sharp_kernel<RP, IXY, CAPA_, LIM_, K>), grid, dim3(256), 0, l.stream, a, ntiles_across, \
                       ntiles_along)
    if (a.src_id != 0) {
        // deltaq += dq_src inside the last pass: built for the shock-bubble configuration only
        if constexpr (std::is_same<RP, Euler5>::value && IXY == 2 && K == 3) {
            if (a.src_id != 1 || capa || l.lim_type != 2) { err = "SharpClaw: the fused dq source needs euler_5wave_2d, WENO5 (lim_type 2), no capacity function"; return PCL_EINVAL; }
            hipLaunchKernelGGL((sharp_kernel<RP, IXY, false, 2, K, true>), grid, dim3(256), 0, l.stream, a, ntiles_across,
                               ntiles_along);
            hipError_t e = hipGetLastError();
            return e == hipSuccess ? PCL_OK : hip_fail(err, "sharp launch", e);
        } else {
            err = "SharpClaw: the fused dq source needs euler_5wave_2d, WENO5 (lim_type 2), no capacity function";
            return PCL_EINVAL;
        }
    }
    if constexpr (K > 3) {
        if (l.lim_type != 2) { err = "SharpClaw: weno_order > 5 needs lim_type 2"; return PCL_EINVAL; }
        if (capa) PCL_SHARP_LAUNCH(true, 2); else PCL_SHARP_LAUNCH(false, 2);
    } else {
        if (l.lim_type == 1) { if (capa) PCL_SHARP_LAUNCH(true, 1); else PCL_SHARP_LAUNCH(false, 1); }
        else if (l.lim_type == 2) { if (capa) PCL_SHARP_LAUNCH(true, 2); else PCL_SHARP_LAUNCH(false, 2); }
        else if (l.lim_type == 3) { if (capa) PCL_SHARP_LAUNCH(true, 3); else PCL_SHARP_LAUNCH(false, 3); }
        else { err = "SharpClaw: lim_type must be 1 (tvd2), 2 (WENO5) or 3 (legacy WENO5)"; return PCL_EINVAL; }
    }
#undef PCL_SHARP_LAUNCH
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sharp launch", e);
}
// mbc = (weno_order+1)/2 (sharpclaw.py:479).  HIGH = this solver also has the kernels of weno_order 7..17 (mbc 4..9):
// the 1-D solvers, 2-D advection, acoustics and Euler; the others are built for mbc = 3 only.
template <class RP, int IXY, bool HIGH = false> int launch_sharp_t(const SweepLaunch &l, std::string &err) {
    if (l.a.mbc == 3) return launch_sharp_k<RP, IXY, 3>(l, err);
    if constexpr (HIGH) {
        switch (l.a.mbc) {
        case 4: return launch_sharp_k<RP, IXY, 4>(l, err);
        case 5: return launch_sharp_k<RP, IXY, 5>(l, err);
        case 6: return launch_sharp_k<RP, IXY, 6>(l, err);
        case 7: return launch_sharp_k<RP, IXY, 7>(l, err);
        case 8: return launch_sharp_k<RP, IXY, 8>(l, err);
        case 9: return launch_sharp_k<RP, IXY, 9>(l, err);
        }
        err = "SharpClaw: mbc must be 3..9 (weno_order 5..17)";
        return PCL_EINVAL;
    }
    err = "SharpClaw: this Riemann solver is built for weno_order 5 (mbc 3) only; orders 7..17 exist for the 1-D solvers "
          "and for advection_2d, acoustics_2d, euler_5wave_2d";
    return PCL_EINVAL;
}
}  // namespace

namespace {
template <class RP> int launch_sharp1w(const SweepLaunch &l, std::string &err) {
    const SweepArgs &a = l.a;
    const int nstrips = (a.mx + sstrip(3) - 1) / sstrip(3);
    const dim3 grid((unsigned)((nstrips + 3) / 4));
    if (l.lim_type == 1) hipLaunchKernelGGL((sharp1w_kernel<RP, 1>), grid, dim3(256), 0, l.stream, a, nstrips);
    else hipLaunchKernelGGL((sharp1w_kernel<RP, 2>), grid, dim3(256), 0, l.stream, a, nstrips);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "sharp (wave-based) launch", e);
}
}  // namespace

int launch_sharp(const SweepLaunch &l, std::string &err) {
    const int rp = l.rp;
    if (l.char_decomp == 1) {
        // 1d/sharpclaw/flux1.f90:80-107 (the 2-D flux1.f90 calls rpn2 with a wrong argument list there and cannot run)
        if (l.ndim != 1 || l.a.mbc != 3 || (l.lim_type != 1 && l.lim_type != 2) || l.a.mcapa > 0 || l.a.src_id != 0) {
            err = "SharpClaw char_decomp = 1: 1-D, lim_type 1 (tvd2_wave) or 2 (weno5_wave), mbc 3, no capacity function";
            return PCL_EINVAL;
        }
        if (rp == PCL_RP_ADVECTION_1D) return launch_sharp1w<Advection1D>(l, err);
        if (rp == PCL_RP_ACOUSTICS_1D) return launch_sharp1w<Acoustics1D>(l, err);
        if (rp == PCL_RP_BURGERS_1D) return launch_sharp1w<Burgers1D>(l, err);
        if (rp == PCL_RP_EULER_1D) return launch_sharp1w<Euler1D>(l, err);
        if (rp == PCL_RP_SHALLOW_1D) return launch_sharp1w<Shallow1D>(l, err);
        err = "SharpClaw char_decomp = 1: Riemann solvers without aux arrays (advection, acoustics, Burgers, Euler, shallow water in 1-D)";
        return PCL_EINVAL;
    }
    if (l.ndim == 1) {
        if (rp == PCL_RP_ADVECTION_1D) return launch_sharp_t<Advection1D, 1, true>(l, err);
        if (rp == PCL_RP_ACOUSTICS_1D) return launch_sharp_t<Acoustics1D, 1, true>(l, err);
        if (rp == PCL_RP_BURGERS_1D) return launch_sharp_t<Burgers1D, 1, true>(l, err);
        if (rp == PCL_RP_EULER_1D) return launch_sharp_t<Euler1D, 1, true>(l, err);
        if (rp == PCL_RP_SHALLOW_1D) return launch_sharp_t<Shallow1D, 1, true>(l, err);
        if (rp == PCL_RP_ADVECTION_COLOR_1D) return launch_sharp_t<AdvectionColor1D, 1, true>(l, err);
    } else if (l.ids == 1) {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch_sharp_t<Acoustics2D, 1, true>(l, err);
        if (rp == PCL_RP_ADVECTION_2D) return launch_sharp_t<Advection2D, 1, true>(l, err);
        if (rp == PCL_RP_SHALLOW_2D) return launch_sharp_t<Shallow2D, 1>(l, err);
        if (rp == PCL_RP_VC_ACOUSTICS_2D) return launch_sharp_t<VcAcoustics2D, 1>(l, err);
        if (rp == PCL_RP_VC_ADVECTION_2D) return launch_sharp_t<VcAdvection2D, 1>(l, err);
        if (rp == PCL_RP_SHALLOW_SPHERE_2D) return launch_sharp_t<ShallowSphere, 1>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch_sharp_t<Euler5, 1, true>(l, err);
    } else {
        if (rp == PCL_RP_ACOUSTICS_2D) return launch_sharp_t<Acoustics2D, 2, true>(l, err);
        if (rp == PCL_RP_ADVECTION_2D) return launch_sharp_t<Advection2D, 2, true>(l, err);
        if (rp == PCL_RP_SHALLOW_2D) return launch_sharp_t<Shallow2D, 2>(l, err);
        if (rp == PCL_RP_VC_ACOUSTICS_2D) return launch_sharp_t<VcAcoustics2D, 2>(l, err);
        if (rp == PCL_RP_VC_ADVECTION_2D) return launch_sharp_t<VcAdvection2D, 2>(l, err);
        if (rp == PCL_RP_SHALLOW_SPHERE_2D) return launch_sharp_t<ShallowSphere, 2>(l, err);
        if (rp == PCL_RP_EULER5_2D) return launch_sharp_t<Euler5, 2, true>(l, err);
    }
    err = "Riemann solver id does not match the grid dimension";
    return PCL_EINVAL;
}

int launch_rk(const RkLaunch &r, hipStream_t stream, std::string &err) {
    RkOp o;
    o.d = r.d; o.a = r.a; o.b = r.b; o.c = r.c; o.ca = r.ca; o.cb = r.cb; o.cc = r.cc; o.n = r.n; o.op = r.op;
    hipLaunchKernelGGL(rk_kernel, dim3(2048), dim3(256), 0, stream, o);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? PCL_OK : hip_fail(err, "rk launch", e);
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
