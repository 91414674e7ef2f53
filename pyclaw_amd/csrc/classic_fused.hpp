// classic_fused.hpp -- the dimension-split 2-D step (step2ds.f: x sweeps of every row, then y sweeps of every
// column of the x-swept array) in ONE kernel: q moves through HBM once per STEP instead of once per pass.
//
// Reference path: src/fortran/2d/classic/step2ds.f:83-159 (both loops), flux2.f, limiter.f; boundary conditions as
// the x pass of sweep_kernel evaluates them while loading (solver.py:354-452), the app's radial source applied to the
// finished cell while storing (clawpack.py:156-159) -- the same lane_core, the same order of operations, the same bits
// as the two-pass form (classic.hpp), which stays the path for decomposed blocks, capacity functions, aux-carrying
// Riemann solvers and mbc != 2.
//
// Tile: 16 rows x 64 columns of q with a 2-cell halo on every side, all MEQN planes in LDS (Euler: 40 KB, four
// workgroups of 256 threads per CU; the round's first form was 32 x 64 with 512 threads, two per CU -- still a build
// option, see PCL_FUSED_ROWS below).  Four phases, a barrier in front of the y sweeps and one behind them:
//   load   the tile, ghost cells remapped to their boundary-condition source cells (tiles on the frame only);
//   x      wavefront w sweeps the four rows it loaded (2w, 2w+1, 8+2w, 9+2w): one lane = one cell, the row is a 64-lane
//          strip exactly as in the x pass; the 60 inner lanes put the updated cell back IN PLACE (interior columns only:
//          ghost columns are copied through, step2ds.f:141-146).  All 16 rows are swept: the y sweeps below need q* two
//          rows beyond the rows they update;
//   y      a wavefront takes FOUR columns at a time, 16 rows each (lanes 16h .. 16h+15 = the rows of column c+h; the
//          wavefront shifts across those boundaries only feed halo rows): rows 2..13 of the interior columns put the
//          finished cell back in place;
//   store  rows 2..13 x columns 2..61 (interior cells only), 16 bytes per lane.
// HBM traffic per cell and step: (16*64)/(12*60) reads + 1 write of q = 96.9 B for Euler (32 x 64: 88.8 B) instead of
// 2 x 82.5 B.  The halo rows / columns are computed twice (x phase 16/12 of the rows, y phase 16/12 of the lanes), which
// the wave-uniform no-jump shortcut of lane_core makes cheap wherever the gas is undisturbed -- and there a wavefront
// remembers the last undisturbed state it met (NoJumpMemo, classic.hpp): the same state again costs MEQN compares.
// LDS layout: row-major rows of 64 doubles, the column XOR-swizzled with the row (fswz): the x phase (lanes = columns)
// touches 64 consecutive doubles, the y phase (lanes = rows) 32 distinct 8-byte banks per half-wave.
#pragma once
#include "classic.hpp"

namespace pcl {
namespace PCL_NS {

// Tile shape (build-time A/B: -DPCL_FUSED_ROWS / _COLS / _THREADS; tools/fused_ab.sh).  32 x 64 with 512 threads is two
// 80 KB workgroups per CU for Euler; 16 x 64 and 32 x 32 with 256 threads are four of 40 KB (shorter phases, more of them
// in flight, for 17 % / 7 % more halo work).  Same box, 4096^2 shock-bubble state, ms per step: 32 x 64 0.350, 16 x 64
// 0.331, 32 x 32 0.370 (its 256-byte row pieces stream badly: 0.376 even without arithmetic); dense state 0.835 / 0.94 /
// 0.86 -- there the solver runs the two passes anyway (pclaw.hip, form trials).
#ifndef PCL_FUSED_ROWS
#define PCL_FUSED_ROWS 16
#endif
#ifndef PCL_FUSED_COLS
#define PCL_FUSED_COLS 64
#endif
#ifndef PCL_FUSED_THREADS
#define PCL_FUSED_THREADS 256
#endif
constexpr int F_ROWS = PCL_FUSED_ROWS, F_COLS = PCL_FUSED_COLS;
constexpr int F_OWN_R = F_ROWS - 2 * HALO, F_OWN_C = F_COLS - 2 * HALO;
constexpr int F_THREADS = PCL_FUSED_THREADS, F_WAVES = F_THREADS / WAVE;
constexpr int F_RX = WAVE / F_COLS;          // rows of the tile one wavefront sweeps at a time (x sweeps)
constexpr int F_CY = WAVE / F_ROWS;          // columns one wavefront sweeps at a time (y sweeps)
constexpr int F_PPR = F_COLS / 2;            // 16-byte pairs per tile row
constexpr int F_NK = F_ROWS * F_PPR / F_THREADS;        // 16-byte loads per thread and plane
constexpr int F_SPK = (WAVE / F_PPR) / F_RX;            // x sweeps that cover the rows of one such load
constexpr int F_NS = F_NK * F_SPK;                      // x sweeps per wavefront
static_assert(F_COLS * F_RX == WAVE && F_ROWS * F_CY == WAVE && F_OWN_C % 2 == 0, "tile shape");
static_assert(F_ROWS * F_PPR % F_THREADS == 0 && (WAVE / F_PPR) % F_RX == 0, "tile shape");
static_assert(F_NS * F_RX * F_WAVES == F_ROWS && (F_COLS / F_CY) % F_WAVES == 0, "tile shape");

#ifndef PCL_FUSED_NT      /* 1: the tile loads bypass the caches like the two-pass kernels' (A/B) */
#define PCL_FUSED_NT 0
#endif
// first row of the tile wavefront w loads and sweeps in its x sweep k = 0..F_NS-1 (16 x 64, 256 threads: 2w, 2w+1,
// 8+2w, 9+2w -- what the threads taking the tile's 16-byte pairs in order give it); lanes >= F_COLS take the next row
__device__ __forceinline__ int wave_row(int w, int k) {
    return (F_THREADS * (k / F_SPK) + WAVE * w) / F_PPR + (k % F_SPK) * F_RX;
}
// column swizzle of row r: a half-wave of the y sweeps (32 / F_ROWS columns x F_ROWS rows) must hit 32 distinct banks
__device__ __forceinline__ int fswz(int r) { return F_ROWS >= 32 ? r : r * (32 / F_ROWS); }
__device__ __forceinline__ int ftile_at(int m, int r, int c) { return (m * F_ROWS + r) * F_COLS + (c ^ fswz(r)); }

template <class RP, bool FWAVE, bool SRC>
__global__ __launch_bounds__(F_THREADS, F_THREADS == 512 ? (RP::MEQN > 3 ? 2 : 3) : 1024 / F_THREADS) void step2ds_kernel(SweepArgs a, int ntx, int nty) {
    constexpr int MEQN = RP::MEQN;
    static_assert(RP::NAUX == 0, "solvers without aux arrays");
    static_assert(!SRC || MEQN == 5, "fused source: the Euler solver");
    __shared__ __attribute__((aligned(16))) double tile[MEQN * F_ROWS * F_COLS];

    // (the XCD-contiguous order of xcd_logical_block costs this kernel 2-3 % although it saves HBM reads -- PCL_TUNE_XCD
    // bit 1 switches it on for A/B)
    int bid = (a.xcd & 2) ? xcd_logical_block(1) : (int)blockIdx.x;
    if (a.xcd & 4) {
        // chunked order (the default; PCL_TUNE_XCD bit 2): the hardware deals consecutive workgroups to the 8 XCDs in
        // turn; in every window of 64 tiles each XCD takes 8 CONSECUTIVE tiles of a tile row (they share partial lines
        // and halo columns in that XCD's L2) while the windows still walk the grid in row order: without arithmetic
        // 0.288 -> 0.274 ms, the shock-bubble step 0.331 -> 0.330 (16 x 64, before the memo)
        const int nb = gridDim.x, win = bid >> 6;
        if ((win + 1) << 6 <= nb) bid = (win << 6) + ((bid & 7) << 3) + ((bid >> 3) & 7);
    }
    int tx = bid % ntx, ty = bid / ntx;
    if ((a.xcd & 8) && a.sub == 0 && ntx >= 8) {
        // column bands (A/B only, PCL_TUNE_XCD bit 3; measured 25 % SLOWER on the shock-bubble state, equal without
        // arithmetic): XCD x walks the tiles of column band x row by row, so a tile's
        // neighbours to the left, right, above and below run on the same XCD (halo rows / columns and shared partial
        // lines hit its L2) while all eight XCDs advance through the tile rows together.  The contiguous ranges of
        // xcd_logical_block over the band-major order of the tiles: a bijection for any grid (bands differ by one column)
        int l = xcd_logical_block(1);
        const int wlo = ntx >> 3, nwide = ntx & 7;       // bands 0..nwide-1 have wlo + 1 columns
        const int wide = nwide * (wlo + 1) * nty;
        int b, w, r;
        if (l < wide) { w = wlo + 1; b = l / (w * nty); r = l - b * w * nty; tx = b * w; }
        else { l -= wide; w = wlo; b = l / (w * nty); r = l - b * w * nty; tx = nwide * (wlo + 1) + b * w; }
        ty = r / w;
        tx += r - ty * w;
    }
    if (a.sub != 0) {
        // decomposed block (pclaw.hip): the tiles inside box = [ty_lo, ty_hi) x [tx_lo, tx_hi) read no ghost cell a
        // neighbour block has to send -- sub 1 = those (they run beside the halo exchange), 2 = the others
        const int bw = a.box[3] - a.box[2];
        if (a.sub == 1) {
            ty = a.box[0] + bid / bw;
            tx = a.box[2] + bid % bw;
        } else {
            const int top = a.box[0] * ntx, bot = (nty - a.box[1]) * ntx;
            if (bid < top) {
                ty = bid / ntx;
                tx = bid % ntx;
            } else if (bid < top + bot) {
                bid -= top;
                ty = a.box[1] + bid / ntx;
                tx = bid % ntx;
            } else {
                bid -= top + bot;
                const int side = ntx - bw;
                const int k = bid % side;
                ty = a.box[0] + bid / side;
                tx = k < a.box[2] ? k : a.box[3] + (k - a.box[2]);
            }
        }
    }
    const int x0 = a.mbc - HALO + tx * F_OWN_C;      // array column of tile column 0 (a.mbc == HALO: checked by the launcher)
    const int y0 = a.mbc - HALO + ty * F_OWN_R;

    // ---- load ----------------------------------------------------------------------------------------------------
    const bool full_tile = x0 + F_COLS <= a.I && y0 + F_ROWS <= a.J;
    const bool vbc_tile = a.vbc_on && ((x0 < a.mbc && a.vbc[0] >= 0) || (x0 + F_COLS > a.I - a.mbc && a.vbc[1] >= 0) ||
                                       (y0 < a.mbc && a.vbc[2] >= 0) || (y0 + F_ROWS > a.J - a.mbc && a.vbc[3] >= 0));
    if (full_tile && !vbc_tile) {
#pragma unroll
        for (int k = 0; k < F_ROWS * F_COLS / 2 / F_THREADS; k++) {
            const int slot = threadIdx.x + F_THREADS * k;
            const int r = slot / (F_COLS / 2), c = 2 * (slot % (F_COLS / 2));
            const long g = (long)(y0 + r) * a.pitch + (x0 + c);
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
#if PCL_FUSED_NT
                double2 v = ld_stream2(&a.qin[m * a.plane + g]);
#else
                double2 v = *reinterpret_cast<const double2 *>(&a.qin[m * a.plane + g]);
#endif
                if (fswz(r) & 1) { const double t = v.x; v.x = v.y; v.y = t; }       // the swizzle swaps the pair in odd rows
                *reinterpret_cast<double2 *>(&tile[(m * F_ROWS + r) * F_COLS + ((c ^ fswz(r)) & ~1)]) = v;
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < F_ROWS * F_COLS / F_THREADS; k++) {
            // the rows this thread's WAVEFRONT sweeps (wave_row below): the same rows the 16-byte path gives it
            const int r = wave_row(threadIdx.x / WAVE, k) + (threadIdx.x & (WAVE - 1)) / F_COLS, c = threadIdx.x & (F_COLS - 1);
            int gx = x0 + c, gy = y0 + r;
            gx = gx < a.I ? gx : a.I - 1;            // past the edge: repeat the last cell (never feeds a stored value)
            gy = gy < a.J ? gy : a.J - 1;
            if (vbc_tile) {
                // qbc = Y(X(q)): x sides first, then y sides over the x-filled array (solver.py:354-381)
                const VbcMap mi = vbc_map(gx, a.I, a.mbc, a.vbc[0], a.vbc[1]);
                const VbcMap mj = vbc_map(gy, a.J, a.mbc, a.vbc[2], a.vbc[3]);
                const long gs = (long)mj.src * a.pitch + mi.src;
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    double v = a.qin[m * a.plane + gs];
                    if (m == 1) v = mi.neg ? -v : v;
                    const double cx = mi.side ? a.vconst[1][m] : a.vconst[0][m];
                    v = mi.cst ? cx : v;
                    if (m == 2) v = mj.neg ? -v : v;
                    const double cy = mj.side ? a.vconst[3][m] : a.vconst[2][m];
                    v = mj.cst ? cy : v;
                    tile[ftile_at(m, r, c)] = v;
                }
            } else {
                const long g = (long)gy * a.pitch + gx;
#pragma unroll
                for (int m = 0; m < MEQN; m++) tile[ftile_at(m, r, c)] = a.qin[m * a.plane + g];
            }
        }
    }
    // no workgroup barrier here: a wavefront sweeps exactly the four rows it loaded (the LDS operations of one
    // wavefront stay in order; the fence keeps the compiler from moving the reads of other lanes' writes up)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    double cflx = 0.0, cfly = 0.0;

    // ---- x sweeps of the tile's rows (step2ds.f:83-146) --------------------------------------------------------
    {
        const int cl = lane & (F_COLS - 1);           // (F_RX rows per wavefront where the tile is narrower than it)
        const int ca = x0 + cl;
        const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.mx) && cl >= HALO && cl < F_COLS - HALO;
        const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.mx) && cl >= 1;
        NoJumpMemo<MEQN> memo;
#pragma unroll 1
        for (int k = 0; k < F_NS; k++) {
            const int r0 = wave_row(wv, k), r = r0 + lane / F_COLS;
            if (y0 + r0 >= a.J) continue;             // wave-uniform
            double q[MEQN], qn[MEQN];
#pragma unroll
            for (int m = 0; m < MEQN; m++) q[m] = tile[ftile_at(m, r, cl)];
            bool nojump = false;     // wave-uniform: the cells go back as they came, nothing to put back
            if (a.ablate & 1) {      // diagnostic (tools/kbench.py): the kernel's memory traffic without its arithmetic
#pragma unroll
                for (int m = 0; m < MEQN; m++) qn[m] = q[m];
            } else
                nojump = lane_core<RP, 1, false, FWAVE, false>(q, a.dtd, 1.0, cfl_ok && (F_RX == 1 || y0 + r < a.J), a, qn, cflx,
                                                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &memo);
            if (owned && !nojump) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) tile[ftile_at(m, r, cl)] = qn[m];
            }
        }
    }
    __syncthreads();

    // ---- y sweeps of the tile's columns (step2ds.f:150-159), two columns per wavefront --------------------------
    {
        SweepArgs ay = a;
        ay.dtd = a.dtd_t;
        const int rl = lane & (F_ROWS - 1), h = lane / F_ROWS;
        const int gy = y0 + rl;
        const bool row_owned = (gy >= a.mbc) && (gy < a.mbc + a.my) && rl >= HALO && rl < F_ROWS - HALO;
        const bool row_cfl = (gy >= a.mbc) && (gy <= a.mbc + a.my) && rl >= 1;
        NoJumpMemo<MEQN> memo;
#pragma unroll 1
        for (int p = wv; p < F_COLS / F_CY; p += F_WAVES) {
            const int c = F_CY * p + h, gx = x0 + c;
            // columns that hold q*: the tile's own interior columns and every ghost column (copied through by the x
            // sweeps; step2ds sweeps them too and their wave speeds count for the Courant number)
            auto has_qstar = [&](int cc) {
                const int g = x0 + cc;
                return g < a.I && ((cc >= HALO && cc < F_COLS - HALO) || g < a.mbc || g >= a.mbc + a.mx);
            };
            bool any = false;
#pragma unroll
            for (int j = 0; j < F_CY; j++) any = any || has_qstar(F_CY * p + j);
            if (!any) continue;                                            // wave-uniform
            const bool col_ok = has_qstar(c);
            const bool col_int = c >= HALO && c < F_COLS - HALO && gx >= a.mbc && gx < a.mbc + a.mx;
            double q[MEQN], qn[MEQN];
#pragma unroll
            for (int m = 0; m < MEQN; m++) q[m] = tile[ftile_at(m, rl, c)];
            bool nojump = false;
            if (a.ablate & 1) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) qn[m] = q[m];
            } else
                nojump = lane_core<RP, 2, false, FWAVE, false>(q, ay.dtd, 1.0, row_cfl && col_ok, ay, qn, cfly,
                                                               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, &memo);
            if (row_owned && col_int && !nojump) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) tile[ftile_at(m, rl, c)] = qn[m];
            }
        }
    }
    __syncthreads();

    // ---- store: the tile's own interior cells --------------------------------------------------------------------
    const bool all_interior = full_tile && x0 + HALO >= a.mbc && x0 + HALO + F_OWN_C <= a.mbc + a.mx && y0 + HALO >= a.mbc &&
                              y0 + HALO + F_OWN_R <= a.mbc + a.my;
    if (all_interior) {
        constexpr int PAIRS = F_OWN_C / 2, SLOTS = PAIRS * F_OWN_R;
#pragma unroll
        for (int k = 0; k < (SLOTS + F_THREADS - 1) / F_THREADS; k++) {
            const int slot = threadIdx.x + F_THREADS * k;
            if (slot < SLOTS) {
                const int r = HALO + slot / PAIRS, c = HALO + 2 * (slot % PAIRS);
                const long g = (long)(y0 + r) * a.pitch + (x0 + c);
                double2 v[MEQN];
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    v[m] = *reinterpret_cast<const double2 *>(&tile[(m * F_ROWS + r) * F_COLS + ((c ^ fswz(r)) & ~1)]);
                    if (fswz(r) & 1) { const double t = v[m].x; v[m].x = v[m].y; v[m].y = t; }
                }
                if constexpr (SRC) {
                    const double2 rad = *reinterpret_cast<const double2 *>(&a.aux[g]);
                    euler_radial_source(v[0].x, v[1].x, v[2].x, v[3].x, rad.x, a.dt, a.src_p[0], a.src_p[1]);
                    euler_radial_source(v[0].y, v[1].y, v[2].y, v[3].y, rad.y, a.dt, a.src_p[0], a.src_p[1]);
                }
#pragma unroll
                for (int m = 0; m < MEQN; m++) st_stream2(&a.qout[m * a.plane + g], v[m]);
            }
        }
    } else {
#pragma unroll 1
        for (int slot = threadIdx.x; slot < F_OWN_R * F_OWN_C; slot += F_THREADS) {
            const int r = HALO + slot / F_OWN_C, c = HALO + slot % F_OWN_C;
            const int gx = x0 + c, gy = y0 + r;
            if (gx >= a.mbc && gx < a.mbc + a.mx && gy >= a.mbc && gy < a.mbc + a.my) {
                const long g = (long)gy * a.pitch + gx;
                double v[MEQN];
#pragma unroll
                for (int m = 0; m < MEQN; m++) v[m] = tile[ftile_at(m, r, c)];
                if constexpr (SRC) euler_radial_source(v[0], v[1], v[2], v[3], a.aux[g], a.dt, a.src_p[0], a.src_p[1]);
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = v[m];
            }
        }
    }
    // the two passes have their own dt/d: the larger Courant number of the two is the step's (step2ds.f:136,181)
    cfl_publish(a.cfl, dmax(cfl_value<false>(cflx, a.dtd), cfl_value<false>(cfly, a.dtd_t)));
}

}  // namespace PCL_NS
}  // namespace pcl
