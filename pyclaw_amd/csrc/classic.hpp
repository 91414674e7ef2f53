// classic.hpp -- the classic (Clawpack) wave-propagation sweep as gfx950 kernels.
//
// Reference path restated here (operation order preserved):
//   flux2.f:88-145      Riemann solve, Godunov increment, CFL, limiter, cqxx/fadd
//   limiter.f:33-57 + philim.f:19-55
//   step2ds.f:88-158 (x pass) / :168-242 (y pass): q update by flux differencing
//   step1.f:93-138      (1-D variant: different association of the same terms)
//
// Mapping: ONE LANE = ONE CELL of a 64-cell strip along the sweep axis; lane l also owns
// interface l (between cells l-1 and l).  Neighbour data moves with wavefront DPP shifts
// (v_mov_b32_dpp wave_shr:1 / wave_shl:1), never through memory.  A strip needs a
// 2-cell halo on each side, so a wavefront updates lanes 2..61 (60 cells).
//
// Device layout of q: structure-of-arrays planes, q[m][j][i], i fastest, row pitch a
// multiple of 16 doubles: the x pass reads rows straight from HBM fully coalesced; the
// y pass stages a 64-row x 16-column tile through LDS (coalesced 128-B row segments in,
// column strips out) so both passes run the same lane-per-cell core.
#pragma once
#include <hip/hip_runtime.h>
#include "rp.hpp"

namespace pcl {
namespace PCL_NS {

constexpr int WAVE = 64;
constexpr int HALO = 2;                 // cells of stencil reach on each side
constexpr int STRIP = WAVE - 2 * HALO;  // cells updated per wavefront strip

// ---- wavefront neighbour shifts ------------------------------------------------------
// bound_ctrl:0 => the lane without a source (lane 0 / lane 63) reads 0 and no "old" value has
// to be kept alive, so the shift is a single v_mov_b32_dpp per dword.  Those end lanes
// never feed a stored result (only lanes 2..61 are written).
__device__ __forceinline__ double from_left(double x) {  // lane l <- lane l-1 (lane 0 gets 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, true);  // wave_shr:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double from_right(double x) {  // lane l <- lane l+1 (lane 63 gets 0)
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, true);  // wave_shl:1
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <class T> __device__ __forceinline__ T struct_from_left(const T &t) {
    constexpr int N = sizeof(T) / sizeof(double);
    union U { T t; double d[N]; __device__ U() {} } a, b;
    a.t = t;
#pragma unroll
    for (int k = 0; k < N; k++) b.d[k] = from_left(a.d[k]);
    return b.t;
}

// philim.f:19-55
__device__ __forceinline__ double philim(double a, double b, int meth) {
    const double r = fdiv(b, a);
    switch (meth) {
    case 1: return dmax(0.0, dmin(1.0, r));
    case 2: return dmax(dmax(0.0, dmin(1.0, 2.0 * r)), dmin(2.0, r));
    case 3: return fdiv(r + fabs(r), 1.0 + fabs(r));
    case 4: { const double c = (1.0 + r) / 2.0; return dmax(0.0, dmin(dmin(c, 2.0), 2.0 * r)); }
    case 5: return r;
    }
    return 1.0;
}

// cfl word: non-negative doubles order like their bit patterns
__device__ __forceinline__ void cfl_publish(unsigned long long *word, double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = dmax(v, __shfl_xor(v, off, WAVE));
    if ((threadIdx.x & (WAVE - 1)) == 0) {
        const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
        // the word only grows: a (possibly stale) smaller reading just costs an atomic
        const unsigned long long seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (bits > seen) atomicMax(word, bits);
    }
}

// ---- the per-lane core ------------------------------------------------------------------
// q      : this lane's cell
// dtdx_c : dtdx1d of this lane's cell (dt/dx, divided by capa when present)
// capa   : aux(mcapa) of this cell (CAPA only)
// cfl_ok : interface l is one of i=1..mx+1 (counts for the Courant number)
// returns the updated cell in qn; valid for lanes 2..61 when lanes l-2..l+2 hold real cells.
template <class RP, int IXY, bool CAPA, bool FWAVE, bool DIM1>
__device__ __forceinline__ void lane_core(const double (&q)[RP::MEQN], double dtdx_c, double capa,
                                          bool cfl_ok, const SweepArgs &a,
                                          double (&qn)[RP::MEQN], double &cflmax) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES;
    using Cell = typename RP::Cell;

    const Cell cR = RP::template precell<IXY>(q, a.par);
    const Cell cL = struct_from_left(cR);
    const double dtdx_l = CAPA ? from_left(dtdx_c) : dtdx_c;

    double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN];
    RP::template solve<IXY>(cL, cR, a.par, wave, s, amdq, apdq);

    // Courant number, flux2.f:109-117
    if (cfl_ok) {
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++)
            cflmax = dmax(dmax(cflmax, dtdx_c * s[mw]), -dtdx_l * s[mw]);
    }

    double fadd[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) fadd[m] = 0.0;

    if (a.order != 1) {
        // limiter.f:33-57 -- dotl(i) = w(i-1).w(i) here, dotr(i) = dotl(i+1) from the right lane
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) {
            const int lim = a.mthlim[mw];
            if (lim == 0) continue;
            // (the Fortran starts both sums at 0.d0; 0 + x == x up to the sign of a zero)
            double wn = 0.0, dl = 0.0;
            bool first = true;
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (!RP::template nz<IXY>(mw, m)) continue;
                const double w = wave[mw][m];
                const double wl = from_left(w);
                wn = first ? w * w : wn + w * w;
                dl = first ? wl * w : dl + wl * w;
                first = false;
            }
            const double dr = from_right(dl);
            if (wn != 0.0) {
                const double phi = philim(wn, s[mw] > 0.0 ? dl : dr, lim);
#pragma unroll
                for (int m = 0; m < MEQN; m++)
                    if (RP::template nz<IXY>(mw, m)) wave[mw][m] = phi * wave[mw][m];
            }
        }
        // second-order corrections, flux2.f:127-145 (step1.f:121-128 in 1-D)
        const double dtdxave = 0.5 * (dtdx_l + dtdx_c);
        double coef[MWAVES];
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) {
            const double sa = fabs(s[mw]);
            if (DIM1)
                coef[mw] = 0.5 * sa * (1.0 - sa * dtdxave);
            else if (FWAVE)
                coef[mw] = copysign(1.0, s[mw]) * (1.0 - sa * dtdxave);
            else
                coef[mw] = sa * (1.0 - sa * dtdxave);
        }
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            double c = 0.0;
            bool first = true;
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++)
                if (RP::template nz<IXY>(mw, m)) {
                    c = first ? coef[mw] * wave[mw][m] : c + coef[mw] * wave[mw][m];
                    first = false;
                }
            fadd[m] = DIM1 ? c : 0.5 * c;
        }
    }

    // update, step2ds.f:141-157 / step1.f:93-96,136-138
#pragma unroll
    for (int m = 0; m < MEQN; m++) {
        const double amdq_r = from_right(amdq[m]);
        const double fadd_r = from_right(fadd[m]);
        if (DIM1) {
            double t = q[m] - dtdx_c * apdq[m];
            t = t - dtdx_c * amdq_r;
            qn[m] = (a.order != 1) ? t - dtdx_c * (fadd_r - fadd[m]) : t;
        } else {
            double qadd = -(dtdx_c * apdq[m]);  // 0.d0 - dtdx1d(i)*apdq(m,i)
            qadd = qadd - dtdx_c * amdq_r;
            if (CAPA)
                qn[m] = q[m] + qadd - a.dtd * (fadd_r - fadd[m]) / capa;
            else
                qn[m] = q[m] + qadd - a.dtd * (fadd_r - fadd[m]);
        }
    }
}

// ---- x pass: rows straight from HBM --------------------------------------------------------
// grid: one wavefront per (row, strip); 4 wavefronts per workgroup.
// Covers step2ds ids=1 (all rows incl. ghost rows) and the 1-D step (J == 1).
template <class RP, bool CAPA, bool FWAVE, bool DIM1>
__global__ __launch_bounds__(256) void sweep_x_kernel(SweepArgs a, int nstrips) {
    constexpr int MEQN = RP::MEQN;
    const int lane = threadIdx.x & (WAVE - 1);
    const int row = blockIdx.y;
    const int strip = blockIdx.x * (256 / WAVE) + (threadIdx.x / WAVE);
    if (strip >= nstrips) return;  // whole wavefront leaves together
    const int c0 = a.mbc - HALO + strip * STRIP;
    const int c = c0 + lane;
    const int cc = c < a.I ? c : a.I - 1;  // clamp: lanes past the row end recompute the last cell
    const long base = (long)row * a.pitch + cc;

    double q[MEQN], qn[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) q[m] = a.qin[m * a.plane + base];
    double capa = 1.0, dtdx_c = a.dtd;
    if (CAPA) {
        capa = a.aux[(long)(a.mcapa - 1) * a.plane + base];
        dtdx_c = DIM1 ? a.dt / (a.dx * capa) : a.dtd / capa;
    }
    const bool interior = (c >= a.mbc) && (c < a.mbc + a.mx);
    const bool cfl_ok = (c >= a.mbc) && (c <= a.mbc + a.mx) && lane >= 1;
    double cflmax = 0.0;
    lane_core<RP, 1, CAPA, FWAVE, DIM1>(q, dtdx_c, capa, cfl_ok, a, qn, cflmax);

    if (c < a.I) {
        if (interior) {
            if (lane >= HALO && lane < WAVE - HALO) {
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + base] = qn[m];
            }
        } else {  // ghost columns keep their value (qnew starts as a copy of qold)
#pragma unroll
            for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + base] = q[m];
        }
    }
    cfl_publish(a.cfl, cflmax);
}

// ---- y pass: 64-row x 16-column tiles through LDS ---------------------------------------------
// step2ds ids=2: every column (ghost columns too), rows 1..my updated, ghost rows copied.
constexpr int YT_COLS = 16;
constexpr int YT_PITCH = YT_COLS + 1;  // +1 double: conflict-free column reads

template <class RP, bool CAPA, bool FWAVE>
__global__ __launch_bounds__(256) void sweep_y_kernel(SweepArgs a, int ntiles_i) {
    constexpr int MEQN = RP::MEQN;
    constexpr int NP = MEQN + 1;  // last plane: capa (only touched when CAPA)
    constexpr int NPA = CAPA ? NP : MEQN;
    __shared__ double tile[NPA][WAVE][YT_PITCH];

    const int ti = blockIdx.x % ntiles_i;
    const int tj = blockIdx.x / ntiles_i;
    const int i0 = ti * YT_COLS;
    const int j0 = a.mbc - HALO + tj * STRIP;

    // load: 16 lanes cover one 128-byte row segment
    {
        const int lc = threadIdx.x % YT_COLS;
        const int lr = threadIdx.x / YT_COLS;  // 0..15
        const int i = i0 + lc;
        const int ic = i < a.I ? i : a.I - 1;
#pragma unroll
        for (int rr = 0; rr < WAVE; rr += 256 / YT_COLS) {
            const int r = rr + lr;
            const int j = j0 + r;
            const int jc = j < a.J ? j : a.J - 1;
            const long g = (long)jc * a.pitch + ic;
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[m][r][lc] = a.qin[m * a.plane + g];
            if constexpr (CAPA) tile[NPA - 1][r][lc] = a.aux[(long)(a.mcapa - 1) * a.plane + g];
        }
    }
    __syncthreads();

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    const int j = j0 + lane;
    const bool interior = (j >= a.mbc) && (j < a.mbc + a.my);
    const bool cfl_row = (j >= a.mbc) && (j <= a.mbc + a.my) && lane >= 1;
    double cflmax = 0.0;
    for (int col = wv; col < YT_COLS; col += 256 / WAVE) {
        if (i0 + col >= a.I) break;  // wave-uniform
        double q[MEQN], qn[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++) q[m] = tile[m][lane][col];
        double capa = 1.0, dtdx_c = a.dtd;
        if constexpr (CAPA) { capa = tile[NPA - 1][lane][col]; dtdx_c = a.dtd / capa; }
        lane_core<RP, 2, CAPA, FWAVE, false>(q, dtdx_c, capa, cfl_row, a, qn, cflmax);
        // only this lane ever reads tile[.][lane][col]: update in place
        if (interior && lane >= HALO && lane < WAVE - HALO) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[m][lane][col] = qn[m];
        }
    }
    __syncthreads();

    // store: interior rows owned by this tile (lanes 2..61) + ghost rows (copied through)
    {
        const int lc = threadIdx.x % YT_COLS;
        const int lr = threadIdx.x / YT_COLS;
        const int i = i0 + lc;
#pragma unroll
        for (int rr = 0; rr < WAVE; rr += 256 / YT_COLS) {
            const int r = rr + lr;
            const int jj = j0 + r;
            const bool inner = (jj >= a.mbc) && (jj < a.mbc + a.my);
            const bool mine = inner ? (r >= HALO && r < WAVE - HALO) : true;
            if (i < a.I && jj < a.J && mine) {
                const long g = (long)jj * a.pitch + i;
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = tile[m][r][lc];
            }
        }
    }
    cfl_publish(a.cfl, cflmax);
}

}  // namespace PCL_NS
}  // namespace pcl
