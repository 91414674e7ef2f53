// classic3.hpp -- the UNSPLIT 3-D classic algorithm (step3.f + the transverse part of flux3.f) as gfx950 kernels.
//
// Reference path restated (operation order preserved; oracle: oracle/classic_oracle.c flux3_full / orc_step3):
//   flux3.f:168-258   normal solve, Godunov increment, CFL, limiter, cqxx / fadd        (as the dim-split sweep)
//   flux3.f:260-593   rpt3 x4 (+4 for the correction waves), rptt3 x8, gadd / hadd accumulation
//   step3.f:114-592   x, y, z sweeps over slices 0..m+1; every slice updates the 3 x 3 cells around it
//
// Two kernels per direction.  slices3_kernel: one lane = one cell of a 64-cell strip (like sweep3_kernel); it leaves
// the slice's pieces for each cell -- qadd, fadd(i+1)-fadd(i), gadd(2,-1:1), hadd(2,-1:1): 14 values per component --
// in scratch planes.  combine3_kernel: every interior cell gathers the nine slices around it IN THE ORDER the
// Fortran's loop nest visits them (x and z sweeps: z-like index outer, y-like inner; y sweep: y-like outer) and
// applies their updates with the reference's association.  All three directions read the same qold, and the
// x, y, z contributions are added in that order, exactly like step3.f.  (Throughput was not the aim of this first
// device version: 14 scratch values per component and direction move ~20x the algorithmic bytes.)
//
// Direction roles (step3.f:176,303,470): sweep DIR, y-like = DIR+1, z-like = DIR+2 (cyclic).  aux block of a cell:
// blk[oe+1][of+1][k] = aux component k of the neighbour at y-like offset oe, z-like offset of.
#pragma once
#include "classic.hpp"

namespace pcl {
namespace PCL_NS {

struct Slices3Args {
    double *scr[14];     // [0] qadd, [1] fadd(i+1)-fadd(i), [2+3*(side-1)+(slice+1)] gadd, [8+...] hadd; MEQN planes each
    long s_e, s_f;       // strides (doubles) of the y-like and z-like directions
    int n_e, n_f;        // extents with ghost cells
    int lo_e, hi_e, lo_f, hi_f;   // swept slices: 0..m+1 in both transverse directions (ghost-offset indices)
    int m3, m4;          // method(3) = 10*m3 + m4
    double dty, dtz;     // dt / d(y-like), dt / d(z-like)
};

template <class RP, int DIR>
__device__ __forceinline__ void load_blk(const SweepArgs &a, const Slices3Args &t, long g, int ce, int cf,
                                         double (&blk)[3][3][RP::NAUX]) {
#pragma unroll
    for (int oe = -1; oe <= 1; oe++)
#pragma unroll
        for (int of = -1; of <= 1; of++) {
            // clamp: the outermost slices' far neighbours do not exist; their results are never used (step3.f sweeps
            // 0..m+1 with mbc = 2, so every USED block lies inside the array)
            const int de = (ce + oe < 0) ? 0 : (ce + oe >= t.n_e ? 0 : oe);
            const int df = (cf + of < 0) ? 0 : (cf + of >= t.n_f ? 0 : of);
#pragma unroll
            for (int k = 0; k < RP::NAUX; k++)
                blk[oe + 1][of + 1][k] = a.aux[aux_idx<RP, DIR>(k) * a.plane + g + de * t.s_e + df * t.s_f];
        }
}

// The pieces one slice leaves for its cells (flux3.f:168-593), in registers: qadd, fadd(i+1)-fadd(i), gadd(side, z-like
// offset), hadd(side, y-like offset).  q / auxv: this lane's cell; blkR: its aux block (load_blk); the left lane's block
// and state arrive by DPP shifts.  Shared by the scratch-plane kernel and the marching kernel below.
template <class RP, int DIR>
__device__ __forceinline__ void slice3_pieces(const double (&q)[RP::MEQN], const double (&auxv)[RP::NAUX],
                                              const double (&blkR)[3][3][RP::NAUX], const SweepArgs &a, const Slices3Args &t,
                                              bool cfl_ok, double &cflmax, double (&qadd)[RP::MEQN], double (&df)[RP::MEQN],
                                              double (&gadd)[2][3][RP::MEQN], double (&hadd)[2][3][RP::MEQN]) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES, NAUX = RP::NAUX;
    using Cell = typename RP::Cell;
    double blkL[3][3][NAUX];
    // cell l-1 (A^- dq): the left lane's block (lane 0 has no interface of its own)
#pragma unroll
    for (int oe = 0; oe < 3; oe++)
#pragma unroll
        for (int of = 0; of < 3; of++)
#pragma unroll
            for (int k = 0; k < NAUX; k++) blkL[oe][of][k] = from_left(blkR[oe][of][k]);
    // reciprocals of the impedance sums the transverse solves divide by (rp.hpp: BlkRcp); the left cell's arrive by shift
    const typename RP::BlkRcp rcR = RP::blk_rcp(blkR);
    typename RP::BlkRcp rcL;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int k = 0; k < 2; k++) {
            rcL.y[r][k].r = from_left(rcR.y[r][k].r);
            rcL.z[r][k].r = from_left(rcR.z[r][k].r);
        }
#pragma unroll
    for (int r = 0; r < 3; r++) {      // the denominators themselves: one add each instead of a shift
        rcL.y[r][0].d = blkL[0][r][0] + blkL[1][r][0]; rcL.y[r][1].d = blkL[1][r][0] + blkL[2][r][0];
        rcL.z[r][0].d = blkL[r][0][0] + blkL[r][1][0]; rcL.z[r][1].d = blkL[r][1][0] + blkL[r][2][0];
    }

    const double d = a.dtd;

    const Cell cR = RP::template precell<DIR>(q, a.par, auxv);
    const Cell cL = struct_from_left(cR);
    double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN];
    RP::template solve<DIR>(cL, cR, a.par, wave, s, amdq, apdq);
    cfl_accumulate<false, MWAVES>(s, d, d, cfl_ok, cflmax);

    double cq[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) cq[m] = 0.0;
    if (a.order != 1) {
        // limiter.f:33-57 (same code as lane_core, without its wave-uniform skips)
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) {
            const int lim = a.mthlim[mw];
            if (lim == 0) continue;
            double wn = 0.0, dl = 0.0;
            bool first = true;
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (!RP::template nz<DIR>(mw, m)) continue;
                const double w = wave[mw][m], wl = from_left(w);
                wn = first ? w * w : wn + w * w;
                dl = first ? wl * w : dl + wl * w;
                first = false;
            }
            const double dr = from_right(dl);
            if (wn != 0.0) {
                const double phi = philim(wn, s[mw] > 0.0 ? dl : dr, lim);
#pragma unroll
                for (int m = 0; m < MEQN; m++)
                    if (RP::template nz<DIR>(mw, m)) wave[mw][m] = phi * wave[mw][m];
            }
        }
        const double dtdxave = 0.5 * (d + d);
#pragma unroll
        for (int m = 0; m < MEQN; m++) {       // flux3.f:244-247
            double c = 0.0;
            bool first = true;
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++)
                if (RP::template nz<DIR>(mw, m)) {
                    const double sa = fabs(s[mw]);
                    const double term = 0.5 * sa * (1.0 - sa * dtdxave) * wave[mw][m];
                    c = first ? term : c + term;
                    first = false;
                }
            cq[m] = c;
        }
    }
#pragma unroll
    for (int m = 0; m < MEQN; m++) {
        const double amdq_r = from_right(amdq[m]), fadd_r = from_right(cq[m]);
        qadd[m] = -(d * apdq[m]) - d * amdq_r;
        df[m] = fadd_r - cq[m];
    }

#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int m = 0; m < MEQN; m++) { gadd[k][j][m] = 0.0; hadd[k][j][m] = 0.0; }

    if (t.m3 > 0) {
        // ---- transverse splits of the fluctuations (flux3.f:269-291) and of the correction waves (:299-321)
        double bmamdq[MEQN], bpamdq[MEQN], bmapdq[MEQN], bpapdq[MEQN];
        double cmamdq[MEQN], cpamdq[MEQN], cmapdq[MEQN], cpapdq[MEQN];
        RP::template transverse3<DIR>(2, blkL, rcL, amdq, bmamdq, bpamdq);
        RP::template transverse3<DIR>(2, blkR, rcR, apdq, bmapdq, bpapdq);
        RP::template transverse3<DIR>(3, blkL, rcL, amdq, cmamdq, cpamdq);
        RP::template transverse3<DIR>(3, blkR, rcR, apdq, cmapdq, cpapdq);
        double bmcqxxm[MEQN], bpcqxxm[MEQN], bmcqxxp[MEQN], bpcqxxp[MEQN];
        double cmcqxxm[MEQN], cpcqxxm[MEQN], cmcqxxp[MEQN], cpcqxxp[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++)
            bmcqxxm[m] = bpcqxxm[m] = bmcqxxp[m] = bpcqxxp[m] = cmcqxxm[m] = cpcqxxm[m] = cmcqxxp[m] = cpcqxxp[m] = 0.0;
        if (t.m3 == 2) {
            RP::template transverse3<DIR>(2, blkL, rcL, cq, bmcqxxm, bpcqxxm);
            RP::template transverse3<DIR>(2, blkR, rcR, cq, bmcqxxp, bpcqxxp);
            RP::template transverse3<DIR>(3, blkL, rcL, cq, cmcqxxm, cpcqxxm);
            RP::template transverse3<DIR>(3, blkR, rcR, cq, cmcqxxp, cpcqxxp);
        }
        const double k6z = (1.0 / 6.0) * d * t.dtz, k6y = (1.0 / 6.0) * d * t.dty;
        double bmcpapdq[MEQN], bpcpapdq[MEQN], bmcpamdq[MEQN], bpcpamdq[MEQN];
        double bmcmapdq[MEQN], bpcmapdq[MEQN], bmcmamdq[MEQN], bpcmamdq[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++)
            bmcpapdq[m] = bpcpapdq[m] = bmcpamdq[m] = bpcpamdq[m] = bmcmapdq[m] = bpcmapdq[m] = bmcmamdq[m] = bpcmamdq[m] = 0.0;

        // ---- G fluxes (y-like), flux3.f:347-452
        if (t.m4 > 0) {
            double cpapdq2[MEQN], cpamdq2[MEQN], cmapdq2[MEQN], cmamdq2[MEQN];
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (t.m4 == 2) {
                    cpapdq2[m] = cpapdq[m] - 3.0 * cpcqxxp[m];
                    cpamdq2[m] = cpamdq[m] + 3.0 * cpcqxxm[m];
                    cmapdq2[m] = cmapdq[m] - 3.0 * cmcqxxp[m];
                    cmamdq2[m] = cmamdq[m] + 3.0 * cmcqxxm[m];
                } else {
                    cpapdq2[m] = cpapdq[m]; cpamdq2[m] = cpamdq[m]; cmapdq2[m] = cmapdq[m]; cmamdq2[m] = cmamdq[m];
                }
            }
            RP::template transverse3t<DIR>(2, 2, blkR, rcR, cpapdq2, bmcpapdq, bpcpapdq);
            RP::template transverse3t<DIR>(2, 2, blkL, rcL, cpamdq2, bmcpamdq, bpcpamdq);
            RP::template transverse3t<DIR>(2, 1, blkR, rcR, cmapdq2, bmcmapdq, bpcmapdq);
            RP::template transverse3t<DIR>(2, 1, blkL, rcL, cmamdq2, bmcmamdq, bpcmamdq);
        }
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            // the A^- parts of interface l+1 belong to this cell: they arrive from the right-hand lane
            const double r_bmamdq = from_right(bmamdq[m]), r_bpamdq = from_right(bpamdq[m]);
            const double r_bmcpamdq = from_right(bmcpamdq[m]), r_bpcpamdq = from_right(bpcpamdq[m]);
            const double r_bmcmamdq = from_right(bmcmamdq[m]), r_bpcmamdq = from_right(bpcmamdq[m]);
            const double r_bmcqxxm = from_right(bmcqxxm[m]), r_bpcqxxm = from_right(bpcqxxm[m]);
            double g10 = 0.0, g20 = 0.0, g21 = 0.0, g11 = 0.0, g2m = 0.0, g1m = 0.0;
            // iteration i = l of flux3.f's loop 180 (index i)
            g10 = g10 - 0.5 * d * bmapdq[m];
            g20 = g20 - 0.5 * d * bpapdq[m];
            if (t.m4 > 0) {
                g20 = g20 + k6z * (bpcpapdq[m] - bpcmapdq[m]);
                g10 = g10 + k6z * (bmcpapdq[m] - bmcmapdq[m]);
                g21 = g21 - k6z * bpcpapdq[m];
                g11 = g11 - k6z * bmcpapdq[m];
                g2m = g2m + k6z * bpcmapdq[m];
                g1m = g1m + k6z * bmcmapdq[m];
            }
            if (t.m3 >= 2) {
                g20 = g20 + d * bpcqxxp[m];
                g10 = g10 + d * bmcqxxp[m];
            }
            // iteration i = l+1 (index i-1)
            g10 = g10 - 0.5 * d * r_bmamdq;
            g20 = g20 - 0.5 * d * r_bpamdq;
            if (t.m4 > 0) {
                g20 = g20 + k6z * (r_bpcpamdq - r_bpcmamdq);
                g10 = g10 + k6z * (r_bmcpamdq - r_bmcmamdq);
                g21 = g21 - k6z * r_bpcpamdq;
                g11 = g11 - k6z * r_bmcpamdq;
                g2m = g2m + k6z * r_bpcmamdq;
                g1m = g1m + k6z * r_bmcmamdq;
            }
            if (t.m3 >= 2) {
                g20 = g20 - d * r_bpcqxxm;
                g10 = g10 - d * r_bmcqxxm;
            }
            gadd[0][1][m] = g10; gadd[1][1][m] = g20; gadd[1][2][m] = g21; gadd[0][2][m] = g11;
            gadd[1][0][m] = g2m; gadd[0][0][m] = g1m;
        }
        // ---- H fluxes (z-like), flux3.f:462-590
        if (t.m4 == 2) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                bpapdq[m] = bpapdq[m] - 3.0 * bpcqxxp[m];
                bpamdq[m] = bpamdq[m] + 3.0 * bpcqxxm[m];
                bmapdq[m] = bmapdq[m] - 3.0 * bmcqxxp[m];
                bmamdq[m] = bmamdq[m] + 3.0 * bmcqxxm[m];
            }
        }
        if (t.m4 > 0) {
            RP::template transverse3t<DIR>(3, 2, blkR, rcR, bpapdq, bmcpapdq, bpcpapdq);
            RP::template transverse3t<DIR>(3, 2, blkL, rcL, bpamdq, bmcpamdq, bpcpamdq);
            RP::template transverse3t<DIR>(3, 1, blkR, rcR, bmapdq, bmcmapdq, bpcmapdq);
            RP::template transverse3t<DIR>(3, 1, blkL, rcL, bmamdq, bmcmamdq, bpcmamdq);
        }
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            const double r_cmamdq = from_right(cmamdq[m]), r_cpamdq = from_right(cpamdq[m]);
            const double r_bmcpamdq = from_right(bmcpamdq[m]), r_bpcpamdq = from_right(bpcpamdq[m]);
            const double r_bmcmamdq = from_right(bmcmamdq[m]), r_bpcmamdq = from_right(bpcmamdq[m]);
            const double r_cmcqxxm = from_right(cmcqxxm[m]), r_cpcqxxm = from_right(cpcqxxm[m]);
            double h10 = 0.0, h20 = 0.0, h21 = 0.0, h11 = 0.0, h2m = 0.0, h1m = 0.0;
            h10 = h10 - 0.5 * d * cmapdq[m];
            h20 = h20 - 0.5 * d * cpapdq[m];
            if (t.m4 > 0) {
                h20 = h20 + k6y * (bpcpapdq[m] - bpcmapdq[m]);
                h10 = h10 + k6y * (bmcpapdq[m] - bmcmapdq[m]);
                h21 = h21 - k6y * bpcpapdq[m];
                h11 = h11 - k6y * bmcpapdq[m];
                h2m = h2m + k6y * bpcmapdq[m];
                h1m = h1m + k6y * bmcmapdq[m];
            }
            if (t.m3 >= 2) {
                h20 = h20 + d * cpcqxxp[m];
                h10 = h10 + d * cmcqxxp[m];
            }
            h10 = h10 - 0.5 * d * r_cmamdq;
            h20 = h20 - 0.5 * d * r_cpamdq;
            if (t.m4 > 0) {
                h20 = h20 + k6y * (r_bpcpamdq - r_bpcmamdq);
                h10 = h10 + k6y * (r_bmcpamdq - r_bmcmamdq);
                h21 = h21 - k6y * r_bpcpamdq;
                h11 = h11 - k6y * r_bmcpamdq;
                h2m = h2m + k6y * r_bpcmamdq;
                h1m = h1m + k6y * r_bmcmamdq;
            }
            if (t.m3 >= 2) {
                h20 = h20 - d * r_cpcqxxm;
                h10 = h10 - d * r_cmcqxxm;
            }
            hadd[0][1][m] = h10; hadd[1][1][m] = h20; hadd[1][2][m] = h21; hadd[0][2][m] = h11;
            hadd[1][0][m] = h2m; hadd[0][0][m] = h1m;
        }
    }
}

template <class RP, int DIR>
__global__ __launch_bounds__(256) void slices3_kernel(SweepArgs a, Slices3Args t, int ntiles_al) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES, NAUX = RP::NAUX;
    using Cell = typename RP::Cell;
    // One wavefront = one 64-cell strip along the sweep; a workgroup = 4 strips.  x direction: 4 consecutive y-like
    // rows.  y and z directions: the lanes' accesses are `pitch` apart, so the 4 wavefronts of a workgroup -- and
    // consecutive workgroups, kept on one XCD by xcd_logical_block -- take CONSECUTIVE i: together they use whole
    // 128-byte lines while those are still in that XCD's L2 (with i spread over blockIdx.y every line was fetched
    // from / written to HBM up to 16 times: 54 ms instead of 6 ms per direction at 256^3).
    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    int ta, ce, cf;
    if (DIR == 1) {
        ta = blockIdx.x % ntiles_al;
        ce = (blockIdx.x / ntiles_al) * 4 + wv + t.lo_e;
        cf = blockIdx.y + t.lo_f;
    } else {
        const int bx = xcd_logical_block(a.xcd);
        const int ngrp = gridDim.x / ntiles_al;                // groups of 4 consecutive i
        const int ci = (bx % ngrp) * 4 + wv;
        ta = bx / ngrp;
        if (DIR == 2) { cf = ci + t.lo_f; ce = blockIdx.y + t.lo_e; }   // y sweep: z-like index = i
        else { ce = ci + t.lo_e; cf = blockIdx.y + t.lo_f; }           // z sweep: y-like index = i
    }
    if (ce > t.hi_e || cf > t.hi_f) return;                    // wave-uniform
    const int a0 = a.mbc - HALO + ta * STRIP;
    const int ca = a0 + lane;
    const int cc = ca < a.n_al ? ca : a.n_al - 1;
    const long g = (long)cc * a.s_al + (long)ce * t.s_e + (long)cf * t.s_f;
    double q[MEQN], auxv[NAUX];
#pragma unroll
    for (int m = 0; m < MEQN; m++) q[m] = a.qin[m * a.plane + g];
#pragma unroll
    for (int k = 0; k < NAUX; k++) auxv[k] = a.aux[aux_idx<RP, DIR>(k) * a.plane + g];
    double blkR[3][3][NAUX];
    load_blk<RP, DIR>(a, t, g, ce, cf, blkR);     // cell l: A^+ dq of interface l sits here
    const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.m_al) && lane >= 1;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.m_al) && lane >= HALO && lane < WAVE - HALO;
    double cflmax = 0.0;
    double qadd[MEQN], df[MEQN], gadd[2][3][MEQN], hadd[2][3][MEQN];
    slice3_pieces<RP, DIR>(q, auxv, blkR, a, t, cfl_ok, cflmax, qadd, df, gadd, hadd);
    if (owned) {
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            const long at = m * a.plane + g;
            t.scr[0][at] = qadd[m];
            t.scr[1][at] = df[m];
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    t.scr[2 + 3 * k + j][at] = gadd[k][j][m];
                    t.scr[8 + 3 * k + j][at] = hadd[k][j][m];
                }
        }
    }
    cfl_publish(a.cfl, cfl_value<false>(cflmax, a.dtd));
}

// ---- the same pieces for solvers whose transverse splits are driven by ONE component ---------------------------------
// (RP::T3_PRESSURE: acoustics.)  rpt3 / rptt3_vc_acoustics read asdq(1) and asdq(iuvw+1) (oracle/classic_oracle.c), and
// what they are handed never has the second one: the normal fluctuations carry (p, sweep velocity), a y-like split
// returns (p, y-like velocity), a z-like split (p, z-like velocity).  So all 16 solves per interface reduce to
//     a1 = -t0 / (Zm + Z),  a2 = t0 / (Z + Zp)      outputs  (cm a1 Zm, -cm a1)  and  (cp a2 Zp, cp a2),
// the G terms live in components (p, y-like velocity), the H terms in (p, z-like velocity), qadd / fadd in
// (p, sweep velocity).  This form keeps only those: 24 doubles of gadd / hadd instead of 48, half the work after the
// normal solve, half the exchange.  Every operation that remains is the dense code's, in its order; the ones dropped
// have a structural zero as operand, so the values agree except, possibly, in the SIGN of a zero (v + 0.0 for
// v = -0.0; -t0 + 0.0*Z for t0 = 0) -- the difference DESIGN section 4.1 already accepts for dmax1/dmin1.
#ifndef PCL_T3_SHARED_RCP
#define PCL_T3_SHARED_RCP 0      /* 1: spills at the 256-VGPR budget of 8 wavefronts (140 B of scratch per lane) */
#endif
struct P2 { double p, v; };
// M34: method(3) = 10*m3 + m4 as a compile-time constant (22: the solver default, every term present, no run-time
// selects), or -1 = read at run time
template <class RP, int DIR, int M34 = -1>
__device__ __forceinline__ void slice3_pieces_p(const double (&q)[RP::MEQN], const double (&auxv)[RP::NAUX],
                                                const double (&blkR)[3][3][RP::NAUX], const SweepArgs &a, const Slices3Args &t,
                                                bool cfl_ok, double &cflmax, P2 &qadd, P2 &df, P2 (&G)[2][3], P2 (&H)[2][3]) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES, NAUX = RP::NAUX;
    using Cell = typename RP::Cell;
    const int m3 = M34 >= 0 ? M34 / 10 : t.m3, m4 = M34 >= 0 ? M34 % 10 : t.m4;
    static_assert(NAUX == 2, "aux = (impedance, sound speed)");
    const double d = a.dtd;
    const Cell cR = RP::template precell<DIR>(q, a.par, auxv);
    const Cell cL = struct_from_left(cR);
    double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN];
    RP::template solve<DIR>(cL, cR, a.par, wave, s, amdq, apdq);
    cfl_accumulate<false, MWAVES>(s, d, d, cfl_ok, cflmax);
    double cq[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) cq[m] = 0.0;
    if (a.order != 1) {
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) {
            const int lim = a.mthlim[mw];
            if (lim == 0) continue;
            double wn = 0.0, dl = 0.0;
            bool first = true;
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (!RP::template nz<DIR>(mw, m)) continue;
                const double w = wave[mw][m], wl = from_left(w);
                wn = first ? w * w : wn + w * w;
                dl = first ? wl * w : dl + wl * w;
                first = false;
            }
            const double dr = from_right(dl);
            if (wn != 0.0) {
                const double phi = philim(wn, s[mw] > 0.0 ? dl : dr, lim);
#pragma unroll
                for (int m = 0; m < MEQN; m++)
                    if (RP::template nz<DIR>(mw, m)) wave[mw][m] = phi * wave[mw][m];
            }
        }
        const double dtdxave = 0.5 * (d + d);
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            if (!(m == 0 || m == DIR)) continue;
            double c = 0.0;
            bool first = true;
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++)
                if (RP::template nz<DIR>(mw, m)) {
                    const double sa = fabs(s[mw]);
                    const double term = 0.5 * sa * (1.0 - sa * dtdxave) * wave[mw][m];
                    c = first ? term : c + term;
                    first = false;
                }
            cq[m] = c;
        }
    }
    qadd.p = -(d * apdq[0]) - d * from_right(amdq[0]);
    qadd.v = -(d * apdq[DIR]) - d * from_right(amdq[DIR]);
    df.p = from_right(cq[0]) - cq[0];
    df.v = from_right(cq[DIR]) - cq[DIR];
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
        for (int j = 0; j < 3; j++) { G[k][j].p = 0.0; G[k][j].v = 0.0; H[k][j].p = 0.0; H[k][j].v = 0.0; }
    if (m3 <= 0) return;

    // Every split below uses THIS cell's block: A^+ dq and the correction flux of interface l belong to cell l, and
    // A^- dq / the correction flux of interface l+1 -- which the dense code splits in lane l+1 with the left
    // neighbour's block and then shifts back -- are fetched from the right-hand lane first and split here.  Same
    // operands, same operations, one lane to the left: the left neighbour's block, its reciprocals and 32 shifts of
    // results disappear, and the twelve impedance sums of the block are inverted once.
    const double aP = apdq[0], aM = from_right(amdq[0]);
    const double kP = cq[0], kM = from_right(cq[0]);
    Recip ry[3][2], rz[3][2];
#pragma unroll
    for (int r = 0; r < 3; r++) {
        ry[r][0] = Recip(blkR[0][r][0] + blkR[1][r][0]); ry[r][1] = Recip(blkR[1][r][0] + blkR[2][r][0]);
        rz[r][0] = Recip(blkR[r][0][0] + blkR[r][1][0]); rz[r][1] = Recip(blkR[r][1][0] + blkR[r][2][0]);
    }
    // one split: the line (Zm, Z, Zp; cm, cp) of the block along the y-like (yl) or z-like direction at the other
    // direction's offset r-1, driven by the pressure part t0: minus-going and plus-going result
    auto split = [&](bool yl, int r, double t0, P2 &om, P2 &op) {
        double zm = 0.0, zp = 0.0, cm = 0.0, cp = 0.0;
        Recip by_m, by_p;
#pragma unroll
        for (int x = 0; x < 3; x++) {       // static indices only (r is a constant after inlining)
            if (x == r) {
                zm = yl ? blkR[0][x][0] : blkR[x][0][0]; zp = yl ? blkR[2][x][0] : blkR[x][2][0];
                cm = yl ? blkR[0][x][1] : blkR[x][0][1]; cp = yl ? blkR[2][x][1] : blkR[x][2][1];
                by_m = yl ? ry[x][0] : rz[x][0]; by_p = yl ? ry[x][1] : rz[x][1];
            }
        }
        const double a1 = by_m.div(-t0);
        const double a2 = by_p.div(t0);
        om.p = cm * a1 * zm; om.v = -cm * a1;
        op.p = cp * a2 * zp; op.v = cp * a2;
    };
    const P2 zero{0.0, 0.0};
    // names as in flux3.f; the ...amdq / ...cqxxm ones are those of interface l+1 (the dense code's r_ values)
    P2 bmamdq, bpamdq, bmapdq, bpapdq, cmamdq, cpamdq, cmapdq, cpapdq;
    split(true, 1, aM, bmamdq, bpamdq);
    split(true, 1, aP, bmapdq, bpapdq);
    split(false, 1, aM, cmamdq, cpamdq);
    split(false, 1, aP, cmapdq, cpapdq);
    P2 bmcqxxm = zero, bpcqxxm = zero, bmcqxxp = zero, bpcqxxp = zero, cmcqxxm = zero, cpcqxxm = zero, cmcqxxp = zero, cpcqxxp = zero;
    if (m3 == 2) {
        split(true, 1, kM, bmcqxxm, bpcqxxm);
        split(true, 1, kP, bmcqxxp, bpcqxxp);
        split(false, 1, kM, cmcqxxm, cpcqxxm);
        split(false, 1, kP, cmcqxxp, cpcqxxp);
    }
    const double k6z = (1.0 / 6.0) * d * t.dtz, k6y = (1.0 / 6.0) * d * t.dty;
    // one component of the six G (or H) values of this cell: the dense code's statements in their order.
    // b?a?dq: first-level split in the flux's own direction; x???: second-level results; q???: correction-wave splits
    auto six = [&](double k6, double bmap, double bpap, double bmam, double bpam, double xmcpap, double xpcpap, double xmcmap,
                   double xpcmap, double xmcpam, double xpcpam, double xmcmam, double xpcmam, double qmp, double qpp, double qmm,
                   double qpm, double &o10, double &o20, double &o21, double &o11, double &o2m, double &o1m) {
        double g10 = 0.0, g20 = 0.0, g21 = 0.0, g11 = 0.0, g2m = 0.0, g1m = 0.0;
        g10 = g10 - 0.5 * d * bmap;
        g20 = g20 - 0.5 * d * bpap;
        if (m4 > 0) {
            g20 = g20 + k6 * (xpcpap - xpcmap);
            g10 = g10 + k6 * (xmcpap - xmcmap);
            g21 = g21 - k6 * xpcpap;
            g11 = g11 - k6 * xmcpap;
            g2m = g2m + k6 * xpcmap;
            g1m = g1m + k6 * xmcmap;
        }
        if (m3 >= 2) {
            g20 = g20 + d * qpp;
            g10 = g10 + d * qmp;
        }
        g10 = g10 - 0.5 * d * bmam;                 // interface l+1
        g20 = g20 - 0.5 * d * bpam;
        if (m4 > 0) {
            g20 = g20 + k6 * (xpcpam - xpcmam);
            g10 = g10 + k6 * (xmcpam - xmcmam);
            g21 = g21 - k6 * xpcpam;
            g11 = g11 - k6 * xmcpam;
            g2m = g2m + k6 * xpcmam;
            g1m = g1m + k6 * xmcmam;
        }
        if (m3 >= 2) {
            g20 = g20 - d * qpm;
            g10 = g10 - d * qmm;
        }
        o10 = g10; o20 = g20; o21 = g21; o11 = g11; o2m = g2m; o1m = g1m;
    };
    {   // ---- G fluxes (y-like), flux3.f:347-452: the z-like splits (corrected by the correction waves' for m4 = 2)
        // split again in the y-like direction, inside the z-like row they went to
        P2 bmcpapdq = zero, bpcpapdq = zero, bmcpamdq = zero, bpcpamdq = zero, bmcmapdq = zero, bpcmapdq = zero, bmcmamdq = zero, bpcmamdq = zero;
        if (m4 > 0) {
            const double cpapdq2 = m4 == 2 ? cpapdq.p - 3.0 * cpcqxxp.p : cpapdq.p;
            const double cpamdq2 = m4 == 2 ? cpamdq.p + 3.0 * cpcqxxm.p : cpamdq.p;
            const double cmapdq2 = m4 == 2 ? cmapdq.p - 3.0 * cmcqxxp.p : cmapdq.p;
            const double cmamdq2 = m4 == 2 ? cmamdq.p + 3.0 * cmcqxxm.p : cmamdq.p;
            split(true, 2, cpapdq2, bmcpapdq, bpcpapdq);
            split(true, 2, cpamdq2, bmcpamdq, bpcpamdq);
            split(true, 0, cmapdq2, bmcmapdq, bpcmapdq);
            split(true, 0, cmamdq2, bmcmamdq, bpcmamdq);
        }
        six(k6z, bmapdq.p, bpapdq.p, bmamdq.p, bpamdq.p, bmcpapdq.p, bpcpapdq.p, bmcmapdq.p, bpcmapdq.p, bmcpamdq.p, bpcpamdq.p,
            bmcmamdq.p, bpcmamdq.p, bmcqxxp.p, bpcqxxp.p, bmcqxxm.p, bpcqxxm.p,
            G[0][1].p, G[1][1].p, G[1][2].p, G[0][2].p, G[1][0].p, G[0][0].p);
        six(k6z, bmapdq.v, bpapdq.v, bmamdq.v, bpamdq.v, bmcpapdq.v, bpcpapdq.v, bmcmapdq.v, bpcmapdq.v, bmcpamdq.v, bpcpamdq.v,
            bmcmamdq.v, bpcmamdq.v, bmcqxxp.v, bpcqxxp.v, bmcqxxm.v, bpcqxxm.v,
            G[0][1].v, G[1][1].v, G[1][2].v, G[0][2].v, G[1][0].v, G[0][0].v);
    }
    {   // ---- H fluxes (z-like), flux3.f:462-590: the y-like splits (corrected for m4 = 2) split again in the z-like direction
        P2 ymcpapdq = zero, ypcpapdq = zero, ymcpamdq = zero, ypcpamdq = zero, ymcmapdq = zero, ypcmapdq = zero, ymcmamdq = zero, ypcmamdq = zero;
        if (m4 > 0) {
            const double bpapdq2 = m4 == 2 ? bpapdq.p - 3.0 * bpcqxxp.p : bpapdq.p;
            const double bpamdq2 = m4 == 2 ? bpamdq.p + 3.0 * bpcqxxm.p : bpamdq.p;
            const double bmapdq2 = m4 == 2 ? bmapdq.p - 3.0 * bmcqxxp.p : bmapdq.p;
            const double bmamdq2 = m4 == 2 ? bmamdq.p + 3.0 * bmcqxxm.p : bmamdq.p;
            split(false, 2, bpapdq2, ymcpapdq, ypcpapdq);
            split(false, 2, bpamdq2, ymcpamdq, ypcpamdq);
            split(false, 0, bmapdq2, ymcmapdq, ypcmapdq);
            split(false, 0, bmamdq2, ymcmamdq, ypcmamdq);
        }
        six(k6y, cmapdq.p, cpapdq.p, cmamdq.p, cpamdq.p, ymcpapdq.p, ypcpapdq.p, ymcmapdq.p, ypcmapdq.p, ymcpamdq.p, ypcpamdq.p,
            ymcmamdq.p, ypcmamdq.p, cmcqxxp.p, cpcqxxp.p, cmcqxxm.p, cpcqxxm.p,
            H[0][1].p, H[1][1].p, H[1][2].p, H[0][2].p, H[1][0].p, H[0][0].p);
        six(k6y, cmapdq.v, cpapdq.v, cmamdq.v, cpamdq.v, ymcpapdq.v, ypcpapdq.v, ymcmapdq.v, ypcmapdq.v, ymcpamdq.v, ypcpamdq.v,
            ymcmamdq.v, ypcmamdq.v, cmcqxxp.v, cpcqxxp.v, cmcqxxm.v, cpcqxxm.v,
            H[0][1].v, H[1][1].v, H[1][2].v, H[0][2].v, H[1][0].v, H[0][0].v);
    }
}

// ---- the marching form: no scratch planes ---------------------------------------------------------------------------
// step3.f visits the slices of a direction in a loop nest -- x and z sweeps: z-like index outside, y-like inside; y sweep:
// y-like outside (step3.f:176,303,470) -- and every slice adds to the 3 x 3 cells around it, so a cell receives its nine
// contributions ordered by the OUTER index of the source slice first, the inner one second.  march3_kernel walks the
// outer index ("march axis" M): a workgroup of NW wavefronts holds NW consecutive slices of the inner index ("wave
// axis" W) of one 64-cell strip, and at march step m every wavefront computes the pieces of its slice (w, m) with the
// code of the scratch-plane kernel (slice3_pieces).  A target cell (w, tm) then gets, in the reference's order,
//     from plane m = tm-1:  slices w-1, w, w+1      (first: the accumulator starts from the cell's old value)
//     from plane m = tm  :  slices w-1, w, w+1
//     from plane m = tm+1:  slices w-1, w, w+1      (last: the cell is complete and is stored)
// i.e. three accumulators per wavefront live in registers across march steps (planes m+1, m, m-1) and only the
// contributions to the neighbouring slices w-1 / w+1 cross wavefronts, through LDS: per component 6 targets x 2
// addends (every update is q = (q + A) + B with A the y-like and B the z-like flux term, products and signs applied by
// the SOURCE lane -- the same operations the combine kernel did, so the bits are the same), two components per
// exchange phase: NW * 2 * 12 * 64 doubles = 96 KB for NW = 8.  Tiles overlap by two slices along W (NW/(NW-2)
// recompute) and the march range is cut into segments (one extra source plane at each end) so that a 256^3 grid
// still gives every CU a workgroup.
//     DIR 1: M = z-like (k), W = y-like (j)     DIR 2: M = y-like (k), W = z-like (i)     DIR 3: M = z-like (j), W = y-like (i)
// For the y and z sweeps the wave axis is i: the 8 wavefronts of a workgroup touch 8 neighbouring doubles of every
// line and workgroups that are neighbours along i run on the same XCD (xcd_logical_block), like slices3_kernel.
struct March3Args {
    const double *qsrc;   // what a cell's accumulator starts from: qold (x direction), the state accumulated so far (y, z)
    double *qacc;
    long s_w, s_m;        // strides (doubles) of the wave axis / march axis
    int n_w, n_m, m_w, m_m;   // extents with ghost cells / interior extents
    int seg;              // target planes per workgroup along the march axis
    int ntiles_al, ntiles_w;
};

// the ordered pair of addends slice S gives the cell at (y-like offset oe, z-like offset of) from itself (step3.f's
// update formulas as the combine kernel evaluates them; G_(k,j) = gadd[k-1][j+1], H_(k,j) = hadd[k-1][j+1]):
// q := (q + A) + B.  The centre cell (0,0) has two more addends in front (qadd, -dtd*df), applied by the caller.
template <int MEQN>
__device__ __forceinline__ void pair3(int oe, int of, int m, double dty, double dtz, const double (&gadd)[2][3][MEQN],
                                      const double (&hadd)[2][3][MEQN], double &A, double &B) {
#define G_(k, j) gadd[(k)-1][(j) + 1][m]
#define H_(k, j) hadd[(k)-1][(j) + 1][m]
    if (oe == 0 && of == 0) { A = -(dty * (G_(2, 0) - G_(1, 0))); B = -(dtz * (H_(2, 0) - H_(1, 0))); }
    else if (oe == -1 && of == 0) { A = -(dty * G_(1, 0)); B = -(dtz * (H_(2, -1) - H_(1, -1))); }
    else if (oe == -1 && of == -1) { A = -(dty * G_(1, -1)); B = -(dtz * H_(1, -1)); }
    else if (oe == 0 && of == -1) { A = -(dty * (G_(2, -1) - G_(1, -1))); B = -(dtz * H_(1, 0)); }
    else if (oe == 1 && of == -1) { A = dty * G_(2, -1); B = -(dtz * H_(1, 1)); }
    else if (oe == 1 && of == 0) { A = dty * G_(2, 0); B = -(dtz * (H_(2, 1) - H_(1, 1))); }
    else if (oe == 1 && of == 1) { A = dty * G_(2, 1); B = dtz * H_(2, 1); }
    else if (oe == 0 && of == 1) { A = -(dty * (G_(2, 1) - G_(1, 1))); B = dtz * H_(2, 0); }
    else { A = -(dty * G_(1, 1)); B = dtz * H_(2, -1); }       // (-1, 1)
#undef G_
#undef H_
}

template <class RP, int DIR, int NW>
__global__ __launch_bounds__(NW *WAVE) void march3_kernel(SweepArgs a, Slices3Args t, March3Args g) {
    constexpr int MEQN = RP::MEQN, NAUX = RP::NAUX;
    constexpr bool E_OUTER = DIR == 2;          // the march axis is the y-like index (y sweep), else the z-like one
    constexpr int CP = 2;                        // components per exchange phase
    static_assert(MEQN % CP == 0, "exchange phases take two components");
    // xbuf[w][c][side][o][A|B][lane]: side 0 = for the slice w+1 (W offset +1), 1 = for the slice w-1; o = M offset + 1
    __shared__ double xbuf[NW][CP][2][3][2][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int bid = DIR == 1 ? (int)blockIdx.x : xcd_logical_block(a.xcd);
    // W tiles fastest (neighbours along the wave axis close together), then the strips, then the march segments
    const int tw = bid % g.ntiles_w, ta = (bid / g.ntiles_w) % g.ntiles_al, ts = bid / (g.ntiles_w * g.ntiles_al);
    const int a0 = a.mbc - HALO + ta * STRIP;
    const int ca = a0 + lane;
    const int cc = ca < a.n_al ? ca : a.n_al - 1;
    const int cw = a.mbc - 1 + tw * (NW - 2) + w;                       // this wavefront's slice along the wave axis
    const bool w_live = cw <= a.mbc + g.m_w;                           // slices 0 .. m+1 exist (wave-uniform)
    const int cwc = cw < g.n_w ? cw : g.n_w - 1;
    const bool target_w = w >= 1 && w <= NW - 2 && cw >= a.mbc && cw < a.mbc + g.m_w;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.m_al) && lane >= HALO && lane < WAVE - HALO && target_w;
    const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.m_al) && lane >= 1;
    const int tm0 = a.mbc + ts * g.seg;
    const int tm1 = tm0 + g.seg < a.mbc + g.m_m ? tm0 + g.seg : a.mbc + g.m_m;      // target planes [tm0, tm1)
    const long base = (long)cc * a.s_al + (long)cwc * g.s_w;
    const int wl = w > 0 ? w - 1 : w, wr = w < NW - 1 ? w + 1 : w;       // (the end wavefronts are never targets)
    double cflmax = 0.0;
    double accP[MEQN], acc0[MEQN], accM[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) { accP[m] = 0.0; acc0[m] = 0.0; accM[m] = 0.0; }

    for (int pm = tm0 - 1; pm <= tm1; pm++) {                            // source planes (all within 0 .. m+1)
        const long gc = base + (long)pm * g.s_m;
        double qadd[MEQN], df[MEQN], gadd[2][3][MEQN], hadd[2][3][MEQN];
        if (w_live) {
            double q[MEQN], auxv[NAUX], blkR[3][3][NAUX];
#pragma unroll
            for (int m = 0; m < MEQN; m++) q[m] = a.qin[m * a.plane + gc];
#pragma unroll
            for (int k = 0; k < NAUX; k++) auxv[k] = a.aux[aux_idx<RP, DIR>(k) * a.plane + gc];
            load_blk<RP, DIR>(a, t, gc, E_OUTER ? pm : cwc, E_OUTER ? cwc : pm, blkR);
            slice3_pieces<RP, DIR>(q, auxv, blkR, a, t, cfl_ok, cflmax, qadd, df, gadd, hadd);
        } else {
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                qadd[m] = 0.0; df[m] = 0.0;
#pragma unroll
                for (int k = 0; k < 2; k++)
#pragma unroll
                    for (int j = 0; j < 3; j++) { gadd[k][j][m] = 0.0; hadd[k][j][m] = 0.0; }
            }
        }
        // the plane ahead starts its accumulator from the cell's value (the march never leaves the array: pm+1 <= m+3)
        if (target_w) {      // wave-uniform
#pragma unroll
            for (int m = 0; m < MEQN; m++) accP[m] = g.qsrc[m * a.plane + gc + g.s_m];
        }

#pragma unroll
        for (int ph = 0; ph < MEQN / CP; ph++) {
            // publish what this slice gives the slices w+1 (side 0) and w-1 (side 1) of the planes pm+1, pm, pm-1
#pragma unroll
            for (int c = 0; c < CP; c++) {
                const int m = ph * CP + c;
#pragma unroll
                for (int side = 0; side < 2; side++)
#pragma unroll
                    for (int o = -1; o <= 1; o++) {
                        const int in = side == 0 ? 1 : -1;
                        double A, B;
                        pair3<MEQN>(E_OUTER ? o : in, E_OUTER ? in : o, m, t.dty, t.dtz, gadd, hadd, A, B);
                        xbuf[w][c][side][o + 1][0][lane] = A;
                        xbuf[w][c][side][o + 1][1][lane] = B;
                    }
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < CP; c++) {
                const int m = ph * CP + c;
#pragma unroll
                for (int o = 1; o >= -1; o--) {
                    double v = o == 1 ? accP[m] : (o == 0 ? acc0[m] : accM[m]);
                    v = v + xbuf[wl][c][0][o + 1][0][lane];               // from the slice w-1 (its W offset +1)
                    v = v + xbuf[wl][c][0][o + 1][1][lane];
                    double A, B;
                    pair3<MEQN>(E_OUTER ? o : 0, E_OUTER ? 0 : o, m, t.dty, t.dtz, gadd, hadd, A, B);
                    if (o == 0) { v = v + qadd[m]; v = v - a.dtd * df[m]; }
                    v = v + A;                                             // this slice's own contribution
                    v = v + B;
                    v = v + xbuf[wr][c][1][o + 1][0][lane];               // from the slice w+1 (its W offset -1)
                    v = v + xbuf[wr][c][1][o + 1][1][lane];
                    if (o == 1) accP[m] = v; else if (o == 0) acc0[m] = v; else accM[m] = v;
                }
            }
            __syncthreads();
        }
        // the plane behind is complete
        if (owned && pm - 1 >= tm0 && pm - 1 < tm1) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) g.qacc[m * a.plane + gc - g.s_m] = accM[m];
        }
#pragma unroll
        for (int m = 0; m < MEQN; m++) { accM[m] = acc0[m]; acc0[m] = accP[m]; }
    }
    cfl_publish(a.cfl, cfl_value<false>(cflmax, a.dtd));
}

// the pair of addends of pair3 for the pressure-driven form: A from G (components p, y-like velocity), B from H
// (components p, z-like velocity)
__device__ __forceinline__ void pair3p(int oe, int of, double dty, double dtz, const P2 (&G)[2][3], const P2 (&H)[2][3],
                                       P2 &A, P2 &B) {
#define G_(k, j, c) G[(k)-1][(j) + 1].c
#define H_(k, j, c) H[(k)-1][(j) + 1].c
#define PAIR_(EA, EB) { A.p = EA(p); A.v = EA(v); B.p = EB(p); B.v = EB(v); }
#define A00(c) -(dty * (G_(2, 0, c) - G_(1, 0, c)))
#define B00(c) -(dtz * (H_(2, 0, c) - H_(1, 0, c)))
#define AM0(c) -(dty * G_(1, 0, c))
#define BM0(c) -(dtz * (H_(2, -1, c) - H_(1, -1, c)))
#define AMM(c) -(dty * G_(1, -1, c))
#define BMM(c) -(dtz * H_(1, -1, c))
#define A0M(c) -(dty * (G_(2, -1, c) - G_(1, -1, c)))
#define B0M(c) -(dtz * H_(1, 0, c))
#define APM(c) (dty * G_(2, -1, c))
#define BPM(c) -(dtz * H_(1, 1, c))
#define AP0(c) (dty * G_(2, 0, c))
#define BP0(c) -(dtz * (H_(2, 1, c) - H_(1, 1, c)))
#define APP(c) (dty * G_(2, 1, c))
#define BPP(c) (dtz * H_(2, 1, c))
#define A0P(c) -(dty * (G_(2, 1, c) - G_(1, 1, c)))
#define B0P(c) (dtz * H_(2, 0, c))
#define AMP(c) -(dty * G_(1, 1, c))
#define BMP(c) (dtz * H_(2, -1, c))
    if (oe == 0 && of == 0) PAIR_(A00, B00)
    else if (oe == -1 && of == 0) PAIR_(AM0, BM0)
    else if (oe == -1 && of == -1) PAIR_(AMM, BMM)
    else if (oe == 0 && of == -1) PAIR_(A0M, B0M)
    else if (oe == 1 && of == -1) PAIR_(APM, BPM)
    else if (oe == 1 && of == 0) PAIR_(AP0, BP0)
    else if (oe == 1 && of == 1) PAIR_(APP, BPP)
    else if (oe == 0 && of == 1) PAIR_(A0P, B0P)
    else PAIR_(AMP, BMP)
#undef G_
#undef H_
#undef PAIR_
#undef A00
#undef B00
#undef AM0
#undef BM0
#undef AMM
#undef BMM
#undef A0M
#undef B0M
#undef APM
#undef BPM
#undef AP0
#undef BP0
#undef APP
#undef BPP
#undef A0P
#undef B0P
#undef AMP
#undef BMP
}

// march3_kernel for the pressure-driven form (slice3_pieces_p): component 0 receives both addends of every contribution,
// the y-like velocity only A, the z-like velocity only B, the sweep velocity only the slice's own qadd / fadd -- 24
// doubles per lane cross wavefronts instead of 48, in ONE exchange phase (two barriers per march step).
template <class RP, int DIR, int NW, int M34 = -1>
__global__ __launch_bounds__(NW *WAVE) void march3p_kernel(SweepArgs a, Slices3Args t, March3Args g) {
    constexpr int MEQN = RP::MEQN, NAUX = RP::NAUX;
    static_assert(MEQN == 4 && RP::T3_PRESSURE, "q = (p, u, v, w)");
    constexpr bool E_OUTER = DIR == 2;
    constexpr int IE = DIR % 3 + 1, IF = (DIR + 1) % 3 + 1;      // components of the y-like / z-like velocity
    // xbuf[w][side][o][A.p | B.p | A.v | B.v][lane]: side 0 = for the slice w+1, 1 = for the slice w-1; o = M offset + 1
    __shared__ double xbuf[NW][2][3][4][WAVE];
    // y and z sweeps: the lanes of a wavefront are `pitch` doubles apart in memory, 64 lines per load / store
    // instruction, and with ~18 of them per step the texture addresser -- not the VALU -- set the pace (x direction
    // 1.14 ms, y / z 1.49 ms at 256^3).  Their wave axis is i, the contiguous one: the workgroup loads and stores
    // 64-row x NW-column tiles cooperatively (8 doubles = 64 B per row segment, 8 segments per instruction instead of
    // 64 lines) and transposes them through LDS.  STAGED tiles: the next plane's cell values (tq1), the accumulated
    // state two planes ahead (tq2: what that plane's accumulator starts from), the aux face two planes ahead (ta,
    // NW + 2 columns), and -- in xbuf's memory, free between two exchanges -- the finished plane on its way out.
    constexpr bool STAGED = DIR != 1;
    constexpr int TP = NW + 1, TAP = NW + 3;
    __shared__ double tq1[STAGED ? MEQN : 1][STAGED ? WAVE : 1][STAGED ? TP : 1];
    __shared__ double tq2[STAGED ? MEQN : 1][STAGED ? WAVE : 1][STAGED ? TP : 1];
    __shared__ double tau[STAGED ? NAUX : 1][STAGED ? WAVE : 1][STAGED ? TAP : 1];
    static_assert(!STAGED || MEQN * WAVE * TP <= NW * 2 * 3 * 4 * WAVE, "the outgoing tile lives in xbuf");
    double(*tout)[WAVE][TP] = reinterpret_cast<double(*)[WAVE][TP]>(&xbuf[0][0][0][0][0]);
    constexpr int NT = NW * WAVE;
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int bid = DIR == 1 ? (int)blockIdx.x : xcd_logical_block(a.xcd);
    const int tw = bid % g.ntiles_w, ta = (bid / g.ntiles_w) % g.ntiles_al, ts = bid / (g.ntiles_w * g.ntiles_al);
    const int a0 = a.mbc - HALO + ta * STRIP;
    const int ca = a0 + lane;
    const int cc = ca < a.n_al ? ca : a.n_al - 1;
    const int cw0 = a.mbc - 1 + tw * (NW - 2);
    const int cw = cw0 + w;
    const bool w_live = cw <= a.mbc + g.m_w;                           // wave-uniform
    const int cwc = cw < g.n_w ? cw : g.n_w - 1;
    const bool target_w = w >= 1 && w <= NW - 2 && cw >= a.mbc && cw < a.mbc + g.m_w;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.m_al) && lane >= HALO && lane < WAVE - HALO && target_w;
    const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.m_al) && lane >= 1;
    const int tm0 = a.mbc + ts * g.seg;
    const int tm1 = tm0 + g.seg < a.mbc + g.m_m ? tm0 + g.seg : a.mbc + g.m_m;
    const long base = (long)cc * a.s_al + (long)cwc * g.s_w;
    const int wl = w > 0 ? w - 1 : w, wr = w < NW - 1 ? w + 1 : w;
    // this thread's cell of the cooperative tiles: row tr along the sweep, column tc along the wave axis
    const int tc = threadIdx.x % NW, tr = threadIdx.x / NW;
    const int trow = a0 + tr < a.n_al ? a0 + tr : a.n_al - 1;
    const int tcol = cw0 + tc < g.n_w ? cw0 + tc : g.n_w - 1;
    const long tbase = (long)trow * a.s_al + (long)tcol * g.s_w;
    const bool t_owned = (a0 + tr >= a.mbc) && (a0 + tr < a.mbc + a.m_al) && tr >= HALO && tr < WAVE - HALO && tc >= 1 &&
                         tc <= NW - 2 && cw0 + tc >= a.mbc && cw0 + tc < a.mbc + g.m_w;
    double cflmax = 0.0;
    double accP[MEQN], acc0[MEQN], accM[MEQN], accN[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) { accP[m] = 0.0; acc0[m] = 0.0; accM[m] = 0.0; accN[m] = 0.0; }

    // The 3 x 3 aux block and the cell itself travel with the march: per step only the face ahead (three cells of the
    // wave axis at plane pm+2) and the next cell are needed -- 6 + 4 values instead of 18 + 2 + 4.
    double blkR[3][3][NAUX], qc[MEQN];
    {
        const long g0 = base + (long)(tm0 - 1) * g.s_m;
        load_blk<RP, DIR>(a, t, g0, E_OUTER ? tm0 - 1 : cwc, E_OUTER ? cwc : tm0 - 1, blkR);
#pragma unroll
        for (int m = 0; m < MEQN; m++) qc[m] = a.qin[m * a.plane + g0];
        if (STAGED && target_w) {      // the first target plane's starting value (later ones come through tq2)
#pragma unroll
            for (int m = 0; m < MEQN; m++) accN[m] = g.qsrc[m * a.plane + g0 + g.s_m];
        }
    }
    const long sw_lo = cwc > 0 ? -g.s_w : 0, sw_hi = cwc + 1 < g.n_w ? g.s_w : 0;      // wave-axis neighbours, clamped like load_blk
    for (int pm = tm0 - 1; pm <= tm1; pm++) {
        const long gc = base + (long)pm * g.s_m;
        // ---- loads for the NEXT step, in flight through this step's arithmetic
        double qn[MEQN], face[3][NAUX];
        double gq[MEQN], gs[MEQN], ga[2][NAUX];
        if constexpr (STAGED) {
            const long tg = tbase + (long)pm * g.s_m;
#pragma unroll
            for (int m = 0; m < MEQN; m++) gq[m] = a.qin[m * a.plane + tg + g.s_m];
            if (pm < tm1) {      // uniform
#pragma unroll
                for (int m = 0; m < MEQN; m++) gs[m] = g.qsrc[m * a.plane + tg + 2 * g.s_m];
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int id = threadIdx.x + k2 * NT;
                    const int ac_ = id % (NW + 2), ar_ = id / (NW + 2);
                    int col = cw0 - 1 + ac_;
                    col = col < 0 ? 0 : (col < g.n_w ? col : g.n_w - 1);
                    const int row = a0 + ar_ < a.n_al ? a0 + ar_ : a.n_al - 1;
#pragma unroll
                    for (int k = 0; k < NAUX; k++)
                        ga[k2][k] = ar_ < WAVE ? a.aux[aux_idx<RP, DIR>(k) * a.plane + (long)row * a.s_al + (long)col * g.s_w + (long)(pm + 2) * g.s_m] : 0.0;
                }
            } else {
#pragma unroll
                for (int m = 0; m < MEQN; m++) gs[m] = 0.0;
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++)
#pragma unroll
                    for (int k = 0; k < NAUX; k++) ga[k2][k] = 0.0;
            }
            if (target_w) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) accP[m] = accN[m];      // plane pm+1 starts from the accumulated state
            }
        } else {
#pragma unroll
            for (int m = 0; m < MEQN; m++) qn[m] = a.qin[m * a.plane + gc + g.s_m];
            if (pm < tm1) {      // uniform
#pragma unroll
                for (int k = 0; k < NAUX; k++) {
                    const long at = aux_idx<RP, DIR>(k) * a.plane + gc + 2 * g.s_m;
                    face[0][k] = a.aux[at + sw_lo]; face[1][k] = a.aux[at]; face[2][k] = a.aux[at + sw_hi];
                }
            } else {
#pragma unroll
                for (int k = 0; k < NAUX; k++) { face[0][k] = blkR[1][1][k]; face[1][k] = blkR[1][1][k]; face[2][k] = blkR[1][1][k]; }
            }
            if (target_w) {      // wave-uniform; x direction: the accumulator starts from qold = the next cell itself
#pragma unroll
                for (int m = 0; m < MEQN; m++) accP[m] = qn[m];
            }
        }
        P2 qadd{0.0, 0.0}, df{0.0, 0.0}, G[2][3], H[2][3];
        if (w_live) {
            double auxv[NAUX];
#pragma unroll
            for (int k = 0; k < NAUX; k++) auxv[k] = blkR[1][1][k];
            slice3_pieces_p<RP, DIR, M34>(qc, auxv, blkR, a, t, cfl_ok, cflmax, qadd, df, G, H);
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int j = 0; j < 3; j++) { G[k][j].p = 0.0; G[k][j].v = 0.0; H[k][j].p = 0.0; H[k][j].v = 0.0; }
        }
#pragma unroll
        for (int side = 0; side < 2; side++)
#pragma unroll
            for (int o = -1; o <= 1; o++) {
                const int in = side == 0 ? 1 : -1;
                P2 A, B;
                pair3p(E_OUTER ? o : in, E_OUTER ? in : o, t.dty, t.dtz, G, H, A, B);
                xbuf[w][side][o + 1][0][lane] = A.p;
                xbuf[w][side][o + 1][1][lane] = B.p;
                xbuf[w][side][o + 1][2][lane] = A.v;
                xbuf[w][side][o + 1][3][lane] = B.v;
            }
        __syncthreads();
#pragma unroll
        for (int o = 1; o >= -1; o--) {
            double *acc = o == 1 ? accP : (o == 0 ? acc0 : accM);
            P2 A, B;
            pair3p(E_OUTER ? o : 0, E_OUTER ? 0 : o, t.dty, t.dtz, G, H, A, B);
            double vp = acc[0], ve = acc[IE], vf = acc[IF];
            vp = vp + xbuf[wl][0][o + 1][0][lane];                      // from the slice w-1
            vp = vp + xbuf[wl][0][o + 1][1][lane];
            ve = ve + xbuf[wl][0][o + 1][2][lane];
            vf = vf + xbuf[wl][0][o + 1][3][lane];
            if (o == 0) {
                vp = vp + qadd.p; vp = vp - a.dtd * df.p;
                acc[DIR] = (acc[DIR] + qadd.v) - a.dtd * df.v;
            }
            vp = vp + A.p; vp = vp + B.p;                               // this slice's own
            ve = ve + A.v;
            vf = vf + B.v;
            vp = vp + xbuf[wr][1][o + 1][0][lane];                      // from the slice w+1
            vp = vp + xbuf[wr][1][o + 1][1][lane];
            ve = ve + xbuf[wr][1][o + 1][2][lane];
            vf = vf + xbuf[wr][1][o + 1][3][lane];
            acc[0] = vp; acc[IE] = ve; acc[IF] = vf;
        }
        __syncthreads();
        const bool plane_out = pm - 1 >= tm0 && pm - 1 < tm1;           // the plane behind is complete (uniform)
        if constexpr (STAGED) {
            // transpose: the loaded tiles in, the finished plane out (tout shares xbuf's memory: every wavefront is past
            // the exchange)
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                tq1[m][tr][tc] = gq[m];
                tq2[m][tr][tc] = gs[m];
                tout[m][lane][w] = accM[m];
            }
#pragma unroll
            for (int k2 = 0; k2 < 2; k2++) {
                const int id = threadIdx.x + k2 * NT;
                const int ac_ = id % (NW + 2), ar_ = id / (NW + 2);
                if (ar_ < WAVE) {
#pragma unroll
                    for (int k = 0; k < NAUX; k++) tau[k][ar_][ac_] = ga[k2][k];
                }
            }
            __syncthreads();
#pragma unroll
            for (int m = 0; m < MEQN; m++) { qn[m] = tq1[m][lane][w]; accN[m] = tq2[m][lane][w]; }
#pragma unroll
            for (int k = 0; k < NAUX; k++) {
                if (pm < tm1) { face[0][k] = tau[k][lane][w]; face[1][k] = tau[k][lane][w + 1]; face[2][k] = tau[k][lane][w + 2]; }
                else { face[0][k] = blkR[1][1][k]; face[1][k] = blkR[1][1][k]; face[2][k] = blkR[1][1][k]; }
            }
            if (plane_out && t_owned) {
                const long tg = tbase + (long)(pm - 1) * g.s_m;
#pragma unroll
                for (int m = 0; m < MEQN; m++) g.qacc[m * a.plane + tg] = tout[m][tr][tc];
            }
            __syncthreads();
        } else {
            if (owned && plane_out) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) g.qacc[m * a.plane + gc - g.s_m] = accM[m];
            }
        }
#pragma unroll
        for (int m = 0; m < MEQN; m++) { accM[m] = acc0[m]; acc0[m] = accP[m]; qc[m] = qn[m]; }
        // the block moves one plane along the march axis (y sweep: the y-like index, else the z-like one)
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int k = 0; k < NAUX; k++) {
                if (E_OUTER) { blkR[0][x][k] = blkR[1][x][k]; blkR[1][x][k] = blkR[2][x][k]; blkR[2][x][k] = face[x][k]; }
                else { blkR[x][0][k] = blkR[x][1][k]; blkR[x][1][k] = blkR[x][2][k]; blkR[x][2][k] = face[x][k]; }
            }
    }
    cfl_publish(a.cfl, cfl_value<false>(cflmax, a.dtd));
}

// ghost frame of the accumulated state := the old state's (the first direction writes interior cells only)
__global__ __launch_bounds__(256) void ghost3_copy_kernel(const double *src, double *dst, int meqn, long plane, int I, int J, int K,
                                                           long pitch, int mbc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y, k = blockIdx.z;
    if (i >= I) return;
    if (i >= mbc && i < I - mbc && j >= mbc && j < J - mbc && k >= mbc && k < K - mbc) return;
    const long c = ((long)k * J + j) * pitch + i;
    for (int m = 0; m < meqn; m++) dst[m * plane + c] = src[m * plane + c];
}

// One thread per cell: qacc(T) += the nine slices of direction DIR around T, in the loop order of step3.f
struct Combine3Args {
    const double *scr[14];
    const double *qsrc;   // qold for the x direction (qacc starts as a copy), qacc itself afterwards
    double *qacc;
    long plane, s_al, s_e, s_f;
    int n_al, n_e, n_f, mbc, m_al, m_e, m_f, meqn;
    double dtd, dty, dtz;
    int e_outer;          // 1: the Fortran's outer loop runs over the y-like index (y sweep), 0: over the z-like one
    int first;            // 1: x direction (start from qsrc and write every cell, ghost cells copied through)
    int dir;              // sweep direction 1..3 (thread -> cell mapping)
};
__global__ __launch_bounds__(256) void combine3_kernel(Combine3Args c) {
    // threads run along i (memory-contiguous) whatever the sweep direction: every load and the store are coalesced
    const int pi = blockIdx.x * blockDim.x + threadIdx.x, pj = blockIdx.y, pk = blockIdx.z;
    const int ia = c.dir == 1 ? pi : (c.dir == 2 ? pj : pk);       // index along the sweep
    const int ie = c.dir == 1 ? pj : (c.dir == 2 ? pk : pi);       // y-like
    const int jf = c.dir == 1 ? pk : (c.dir == 2 ? pi : pj);       // z-like
    if (ia >= c.n_al || ie >= c.n_e || jf >= c.n_f) return;
    const long g = (long)ia * c.s_al + (long)ie * c.s_e + (long)jf * c.s_f;
    const bool interior = ia >= c.mbc && ia < c.mbc + c.m_al && ie >= c.mbc && ie < c.mbc + c.m_e && jf >= c.mbc &&
                          jf < c.mbc + c.m_f;
    for (int m = 0; m < c.meqn; m++) {
        const long at = m * c.plane + g;
        double q = c.first ? c.qsrc[at] : c.qacc[at];
        if (interior) {
            for (int o = 1; o >= -1; o--)
                for (int in = 1; in >= -1; in--) {
                    const int oe = c.e_outer ? o : in, of = c.e_outer ? in : o;
                    const long S = at - oe * c.s_e - of * c.s_f;      // the slice that adds to (oe, of) from itself
#define G_(k, j) c.scr[2 + 3 * ((k)-1) + ((j) + 1)][S]
#define H_(k, j) c.scr[8 + 3 * ((k)-1) + ((j) + 1)][S]
                    if (oe == 0 && of == 0)
                        q = q + c.scr[0][S] - c.dtd * c.scr[1][S] - c.dty * (G_(2, 0) - G_(1, 0)) - c.dtz * (H_(2, 0) - H_(1, 0));
                    else if (oe == -1 && of == 0) q = q - c.dty * G_(1, 0) - c.dtz * (H_(2, -1) - H_(1, -1));
                    else if (oe == -1 && of == -1) q = q - c.dty * G_(1, -1) - c.dtz * H_(1, -1);
                    else if (oe == 0 && of == -1) q = q - c.dty * (G_(2, -1) - G_(1, -1)) - c.dtz * H_(1, 0);
                    else if (oe == 1 && of == -1) q = q + c.dty * G_(2, -1) - c.dtz * H_(1, 1);
                    else if (oe == 1 && of == 0) q = q + c.dty * G_(2, 0) - c.dtz * (H_(2, 1) - H_(1, 1));
                    else if (oe == 1 && of == 1) q = q + c.dty * G_(2, 1) + c.dtz * H_(2, 1);
                    else if (oe == 0 && of == 1) q = q - c.dty * (G_(2, 1) - G_(1, 1)) + c.dtz * H_(2, 0);
                    else q = q - c.dty * G_(1, 1) + c.dtz * H_(2, -1);
#undef G_
#undef H_
                }
        }
        if (interior || c.first) c.qacc[at] = q;
    }
}

}  // namespace PCL_NS
}  // namespace pcl
