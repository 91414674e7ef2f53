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
// multiple of 16 doubles, first interior cell of a row on a 128-byte line.  Both passes stage
// tiles through LDS (x: 4 rows x 244 cells; y: 64 rows x 16 columns, coalesced 128-byte row
// segments in, column strips out) and run the same lane-per-cell core (lane_core).
// Contents: lane_core | sweep_kernel (step2ds / step1) | unsplit_x/y_kernel (step2, step2qcor) |
// sweep3_kernel (step3ds); the unsplit 3-D step is classic3.hpp.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

// Streaming access to the state arrays.  Every pass reads each cell of q once (plus a 2-cell halo) and writes it once,
// and the arrays (671 MB per 4096^2 Euler state) are far larger than L2 + the 256 MB Infinity Cache, so nothing a pass
// touches is still cached when the next pass wants it: nontemporal loads / stores stop the lines from displacing each
// other on their way through.  Measured (tools/ubench/copy_rates.hip, 5 planes in / 5 out, this tile shape):
// 6193 GB/s plain, 6853 GB/s nontemporal.  PCL_NT=0 builds the plain form for A/B.
#ifndef PCL_NT
#define PCL_NT 1
#endif
namespace pcl {
namespace PCL_NS {
typedef double pcl_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double ld_stream(const double *p) {
#if PCL_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ double2 ld_stream2(const double *p) {
#if PCL_NT
    const pcl_d2 v = __builtin_nontemporal_load(reinterpret_cast<const pcl_d2 *>(p));
    double2 r; r.x = v.x; r.y = v.y;
    return r;
#else
    return *reinterpret_cast<const double2 *>(p);
#endif
}
__device__ __forceinline__ void st_stream(double *p, double v) {
#if PCL_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void st_stream2(double *p, double2 v) {
#if PCL_NT
    pcl_d2 w; w.x = v.x; w.y = v.y;
    __builtin_nontemporal_store(w, reinterpret_cast<pcl_d2 *>(p));
#else
    *reinterpret_cast<double2 *>(p) = v;
#endif
}
}  // namespace PCL_NS
}  // namespace pcl
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

// ---- which aux components a Riemann solver reads ---------------------------------------------------
// Solvers with cell-wise coefficients name the aux planes they need (rp.hpp): aux_index<IXY>(k), k < NAUX, for
// the sweep itself (staged next to q in the LDS tiles) and auxt_index<IXY>(k), k < NAUX_T, for the transverse
// solves (read for the cell and its two neighbours across the sweep).  Solvers without aux have neither.
template <class RP, int IXY> __host__ __device__ constexpr int aux_idx(int k) {
    if constexpr (RP::NAUX > 0) return RP::template aux_index<IXY>(k);
    else return k;
}
template <class RP, int IXY> __host__ __device__ constexpr int auxt_idx(int k) {
    if constexpr (RP::NAUX > 0) return RP::template auxt_index<IXY>(k);
    else return k;
}
template <class RP> __host__ __device__ constexpr int naux_t() {
    if constexpr (RP::NAUX > 0) return RP::NAUX_T;
    else return 0;
}
// number of aux planes the y phase of the unsplit kernel stages (highest plane index read + 1), and which of them
template <class RP> __host__ __device__ constexpr int aux_planes_y() {
    int n = 0;
    if constexpr (RP::NAUX > 0) {
        for (int k = 0; k < RP::NAUX; k++) n = aux_idx<RP, 2>(k) + 1 > n ? aux_idx<RP, 2>(k) + 1 : n;
        for (int k = 0; k < RP::NAUX_T; k++) n = auxt_idx<RP, 2>(k) + 1 > n ? auxt_idx<RP, 2>(k) + 1 : n;
    }
    return n;
}
template <class RP> __host__ __device__ constexpr bool aux_plane_used_y(int p) {
    bool u = false;
    if constexpr (RP::NAUX > 0) {
        for (int k = 0; k < RP::NAUX; k++) u = u || aux_idx<RP, 2>(k) == p;
        for (int k = 0; k < RP::NAUX_T; k++) u = u || auxt_idx<RP, 2>(k) == p;
    }
    return u;
}
template <class RP, class = void> struct IsFwave : std::false_type {};
template <class RP> struct IsFwave<RP, std::void_t<decltype(RP::IS_FWAVE)>> : std::bool_constant<RP::IS_FWAVE> {};
template <class RP, class = void> struct HasQcor : std::false_type {};
template <class RP> struct HasQcor<RP, std::void_t<decltype(RP::HAS_QCOR)>> : std::bool_constant<RP::HAS_QCOR> {};
template <class RP> __host__ __device__ constexpr bool has_qcor() { return HasQcor<RP>::value; }

// philim.f:19-55
__device__ __forceinline__ double philim(double a, double b, int meth) {
#if PCL_FAST
    // |wave|^2 can underflow (tracer tails): the reciprocal of a denormal is inf and inf*0 a NaN.  By Cauchy-Schwarz
    // |b| <= sqrt(a_neighbour * a), so with a clamped to 1e-300 the ratio stays finite; such a wave carries nothing.
    a = __builtin_fmax(a, 1e-300);
#endif
    const double r = fdiv_ieee(b, a);
    switch (meth) {
    case 1: return dmax(0.0, dmin(1.0, r));
    case 2: return dmax(dmax(0.0, dmin(1.0, 2.0 * r)), dmin(2.0, r));
    case 3: return fdiv_ieee(r + fabs(r), 1.0 + fabs(r));
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

// Courant number of one interface (flux2.f:109-117): max over the waves of dtdx(i)*s and -dtdx(i-1)*s.
// Without a capacity function dtdx is one positive number for the whole grid, and rounding is monotone:
//     max_k fl(d*s_k) = fl(d * max_k s_k)   and   max(fl(d*s), fl(-d*s)) = fl(d*|s|),
// so the running maximum is kept over |s| alone (one v_max_f64 with |.| per wave instead of two multiplies
// and two maxima) and multiplied by dt/dx ONCE per wavefront, in cfl_value() below: the same bits.
// With a capacity function dtdx differs from cell to cell and the reference's expression is kept.
// A NaN speed is ignored either way (v_max_f64 returns the other operand).
template <bool CAPA, int MWAVES>
__device__ __forceinline__ void cfl_accumulate(const double (&s)[MWAVES], double dtdx_c, double dtdx_l, bool cfl_ok,
                                               double &cflmax) {
    if constexpr (CAPA) {
        if (cfl_ok) {
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++) cflmax = dmax(dmax(cflmax, dtdx_c * s[mw]), -dtdx_l * s[mw]);
        }
    } else {
        double m = fabs(s[0]);
#pragma unroll
        for (int mw = 1; mw < MWAVES; mw++) m = dmax(m, fabs(s[mw]));
        if (cfl_ok) cflmax = dmax(cflmax, m);
    }
}
// what a kernel publishes: the accumulated maximum itself (CAPA) or dt/dx times the largest |s|
template <bool CAPA> __device__ __forceinline__ double cfl_value(double cflmax, double dtd) {
    return CAPA ? cflmax : dtd * cflmax;
}

// The last undisturbed state a wavefront met and its largest |wave speed| (one-kernel step, classic_fused.hpp).  In a
// wavefront without a jump every lane holds the SAME cell, so the key is wave-uniform: it lives in scalar registers
// (v_readlane), the test is MEQN compares against scalar operands, and a hit skips precell + speeds (two square
// roots, a handful of quotients) -- in undisturbed gas strip after strip has the same state.  Same bits: equal q gives
// equal speeds (+0 == -0 compares equal and cannot change a |speed|).
template <int MEQN> struct NoJumpMemo {
    double key[MEQN];
    double m;
    int valid = 0;
};
__device__ __forceinline__ double uniform_from_lane1(double x) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 1), hi = __builtin_amdgcn_readlane(__double2hiint(x), 1);
    return __hiloint2double(hi, lo);
}

// ---- the per-lane core ------------------------------------------------------------------
// q      : this lane's cell
// dtdx_c : dtdx1d of this lane's cell (dt/dx, divided by capa when present)
// capa   : aux(mcapa) of this cell (CAPA only)
// cfl_ok : interface l is one of i=1..mx+1 (counts for the Courant number)
// returns the updated cell in qn; valid for lanes 2..61 when lanes l-2..l+2 hold real cells.
//
// TRANS (unsplit algorithm, flux2.f:151-189): instead of the updated cell, return the slice's
// pieces for this cell -- qn := qadd, df := fadd(i+1)-fadd(i), g1/g2 := gadd(.,1,i), gadd(.,2,i)
// (transverse flux corrections from rpt2 of amdq at interface i+1 and of apdq at interface i).
// F3 (3-D, flux3.f:236-252 + step3ds.f:177-181): correction coefficients carry the 0.5 like the 1-D code
// (fadd = cqxx), the update is associated like the 2-D one ((q + qadd) - dtdx*(fadd(i+1)-fadd(i))).
// auxv: this cell's aux values for Riemann solvers with RP::NAUX > 0.
template <class RP, int IXY, bool CAPA, bool FWAVE, bool DIM1, bool TRANS = false, bool F3 = false>
__device__ __forceinline__ bool lane_core(const double (&q)[RP::MEQN], double dtdx_c, double capa,
                                          bool cfl_ok, const SweepArgs &a,
                                          double (&qn)[RP::MEQN], double &cflmax,
                                          double *df = nullptr, double *g1 = nullptr, double *g2 = nullptr,
                                          const double *auxv = nullptr, const double *auxb = nullptr,
                                          const double *auxa = nullptr, const double *auxo = nullptr,
                                          NoJumpMemo<RP::MEQN> *memo = nullptr) {   // returns: the no-jump shortcut was taken (wave-uniform)
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES;
    using Cell = typename RP::Cell;

    const double dtdx_l = CAPA ? from_left(dtdx_c) : dtdx_c;
    double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN];

    // Solvers without aux arrays: the no-jump test (below) on the raw cell values, BEFORE the cell's derived quantities
    // exist -- a jump-free wavefront then needs neither the shift of the whole Cell (12 doubles for Euler) nor the
    // quantities only the full solve reads (the compiler drops them from this branch): equal q means the left cell's
    // Cell equals this lane's bit for bit, so the Roe average is taken of (c, c).  Same speeds, same Courant number.
    if constexpr (!FWAVE && RP::NAUX == 0) {
        const bool lane0 = (threadIdx.x & (WAVE - 1)) == 0;   // lane 0 has no left cell
        // (every shift runs with all lanes active, outside the short-circuit expressions: a DPP read of a lane that is
        // masked off returns nothing useful)
        const double ql0 = from_left(q[0]);
        bool same = !(a.ablate & 16) && __all(lane0 || ql0 == q[0]);
        if (same) {      // wave-uniform
            double ql[MEQN];
#pragma unroll
            for (int m = 1; m < MEQN; m++) ql[m] = from_left(q[m]);
            bool rest = true;
#pragma unroll
            for (int m = 1; m < MEQN; m++) rest = rest & (ql[m] == q[m]);
            same = __all(lane0 || rest);
        }
        if (same) {
            if constexpr (!CAPA && !TRANS) {
                if (memo && memo->valid) {       // wave-uniform: the state of the last jump-free strip again?
                    bool hit = true;
#pragma unroll
                    for (int m = 0; m < MEQN; m++) hit = hit & (q[m] == memo->key[m]);
                    if (__all(hit)) {
                        if (!(a.ablate & 8) && cfl_ok) cflmax = dmax(cflmax, memo->m);
#pragma unroll
                        for (int m = 0; m < MEQN; m++) qn[m] = q[m];
                        return true;
                    }
                }
            }
            const Cell c = RP::template precell<IXY>(q, a.par);
            RP::template speeds<IXY>(c, c, a.par, s);
            bool finite = true;
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++) finite = finite && (s[mw] - s[mw] == 0.0);
            if (__all(finite || lane0)) {
                if (!(a.ablate & 8)) cfl_accumulate<CAPA, MWAVES>(s, dtdx_c, dtdx_l, cfl_ok, cflmax);
                if constexpr (!CAPA && !TRANS) {
                    if (memo) {                  // every lane holds this cell: lane 1's copy is the wavefront's
                        double mx = fabs(s[0]);
#pragma unroll
                        for (int mw = 1; mw < MWAVES; mw++) mx = dmax(mx, fabs(s[mw]));
                        memo->m = uniform_from_lane1(mx);
#pragma unroll
                        for (int m = 0; m < MEQN; m++) memo->key[m] = uniform_from_lane1(q[m]);
                        memo->valid = 1;
                    }
                }
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    if constexpr (TRANS) { qn[m] = 0.0; df[m] = 0.0; g1[m] = 0.0; g2[m] = 0.0; }  // the slice's pieces
                    else qn[m] = q[m];
                }
                return true;
            }
        }
    }
    Cell cR;
    if constexpr (RP::NAUX > 0) cR = RP::template precell<IXY>(q, a.par, auxv);
    else cR = RP::template precell<IXY>(q, a.par);
    const Cell cL = struct_from_left(cR);

    // A wavefront without a single jump (every lane's cell equals its left neighbour: undisturbed gas,
    // the inside of the bubble, a constant post-shock state): all waves, fluctuations and correction
    // fluxes are zero and the update is the identity, whatever the limiter -- the reference computes exactly
    // that (q + 0).  Only the wave speeds are needed, for the Courant number (flux2.f:109-117); they come
    // from the same Roe average the full solve uses.  Wave-uniform branch; a non-finite speed (unphysical
    // state) takes the full path.  PCL_TUNE_ABLATE bit 4 switches the shortcut off (tools/kbench.py).
    if constexpr (!FWAVE && RP::NAUX > 0) {   // f-waves over varying aux are non-zero even for equal q: no shortcut there
        const bool lane0 = (threadIdx.x & (WAVE - 1)) == 0;   // lane 0 has no left cell
        // first component first: where the state varies at all, one compare per lane settles it
        bool same = !(a.ablate & 16) && __all(lane0 || cL.q[0] == cR.q[0]);
        if (same) {
            bool rest = true;
#pragma unroll
            for (int m = 1; m < MEQN; m++) rest = rest && (cL.q[m] == cR.q[m]);
            same = __all(lane0 || rest);
        }
        if (same) {
            RP::template speeds<IXY>(cL, cR, a.par, s);
            bool finite = true;
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++) finite = finite && (s[mw] - s[mw] == 0.0);
            if (__all(finite || (threadIdx.x & (WAVE - 1)) == 0)) {
                if (!(a.ablate & 8)) cfl_accumulate<CAPA, MWAVES>(s, dtdx_c, dtdx_l, cfl_ok, cflmax);
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    if constexpr (TRANS) { qn[m] = 0.0; df[m] = 0.0; g1[m] = 0.0; g2[m] = 0.0; }  // the slice's pieces
                    else qn[m] = q[m];
                }
                return true;
            }
        }
    }
    RP::template solve<IXY>(cL, cR, a.par, wave, s, amdq, apdq);

    // Courant number, flux2.f:109-117
    if (!(a.ablate & 8)) cfl_accumulate<CAPA, MWAVES>(s, dtdx_c, dtdx_l, cfl_ok, cflmax);

    double fadd[MEQN], cq[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) { fadd[m] = 0.0; cq[m] = 0.0; }

    if (a.order != 1 && !(a.ablate & 4)) {
        // limiter.f:33-57 -- dotl(i) = w(i-1).w(i) here, dotr(i) = dotl(i+1) from the right lane.
        // The reference skips an interface whose wave has zero norm (limiter.f:45); when NO lane of the
        // wavefront has a wave in this family (a tracer-free or shear-free stretch, undisturbed gas) the
        // neighbour shifts and dot products are skipped as well (nothing would be limited).
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) {
            const int lim = a.mthlim[mw];
            if (lim == 0 || (a.ablate & 2)) continue;
            double wn = 0.0;
            bool first = true;
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (!RP::template nz<IXY>(mw, m)) continue;
                const double w = wave[mw][m];
                wn = first ? w * w : wn + w * w;   // (the Fortran starts the sum at 0.d0: same value)
                first = false;
            }
            // lane 0 has no left cell (its 'wave' is garbage): it must not keep the family alive
            if (!__any(wn != 0.0 && (threadIdx.x & (WAVE - 1)) != 0)) continue;  // wave-uniform
            double dl = 0.0;
            first = true;
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                if (!RP::template nz<IXY>(mw, m)) continue;
                const double w = wave[mw][m];
                const double wl = from_left(w);
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
            // flux2fw.f:151-152 / step1fw.f:139-140: dsign(1,s) in place of |s| when wave[][] holds f-waves
            const double mag = FWAVE ? copysign(1.0, s[mw]) : sa;
            if (DIM1 || F3)
                coef[mw] = 0.5 * mag * (1.0 - sa * dtdxave);
            else
                coef[mw] = mag * (1.0 - sa * dtdxave);
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
            cq[m] = c;
            fadd[m] = (DIM1 || F3) ? c : 0.5 * c;
        }
    }

    if constexpr (TRANS) {
        // Godunov increment and correction-flux difference of this cell (flux2.f:103-106,144)
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            const double amdq_r = from_right(amdq[m]);
            const double fadd_r = from_right(fadd[m]);
            double qadd = -(dtdx_c * apdq[m]);
            qn[m] = qadd - dtdx_c * amdq_r;
            df[m] = fadd_r - fadd[m];
            g1[m] = 0.0;
            g2[m] = 0.0;
        }
        if (a.trans > 0) {
            if (a.order > 1 && a.trans == 2) {  // flux2.f:153-159: split the correction waves too
#pragma unroll
                for (int m = 0; m < MEQN; m++) { amdq[m] = amdq[m] + cq[m]; apdq[m] = apdq[m] - cq[m]; }
            }
            double bm[MEQN], bp[MEQN];
            // solvers with cell-wise coefficients: transverse aux set of the cell itself and of the cells below /
            // above in the neighbouring slices -- of this lane's cell for A^+ dq, of the left neighbour's
            // (shifted) for A^- dq
            constexpr int NT = naux_t<RP>() > 0 ? naux_t<RP>() : 1;
            double auxol[NT], auxbl[NT], auxal[NT];
            if constexpr (RP::NAUX > 0) {
#pragma unroll
                for (int k = 0; k < NT; k++) {
                    auxol[k] = from_left(auxo[k]); auxbl[k] = from_left(auxb[k]); auxal[k] = from_left(auxa[k]);
                }
            }
            // B^-/B^+ A^- dq of interface l modify the cell to its LEFT (flux2.f:167-176)
            if constexpr (RP::NAUX > 0) RP::template transverse_vc<IXY>(cL, auxol, auxbl, auxal, a.par, amdq, bm, bp);
            else RP::template transverse<IXY>(cL, cR, a.par, amdq, bm, bp);
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                g1[m] = -(0.5 * dtdx_c * from_right(bm[m]));
                g2[m] = -(0.5 * dtdx_c * from_right(bp[m]));
            }
            // B^-/B^+ A^+ dq of interface l modify this cell (flux2.f:180-189)
            if constexpr (RP::NAUX > 0) RP::template transverse_vc<IXY>(cR, auxo, auxb, auxa, a.par, apdq, bm, bp);
            else RP::template transverse<IXY>(cL, cR, a.par, apdq, bm, bp);
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                g1[m] = g1[m] - 0.5 * dtdx_c * bm[m];
                g2[m] = g2[m] - 0.5 * dtdx_c * bp[m];
            }
        }
        return false;
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
    return false;
}

// ---- the sweep kernel: tiles staged through LDS ------------------------------------------------
// One workgroup (4 wavefronts) owns a tile; all 256 threads first pull the tile in with ~20
// independent 8-byte loads each (~40 KB in flight per workgroup, 3-4 workgroups per CU: that
// memory-level parallelism is what lets a pass stream at ~5 TB/s; a wavefront loading only
// its own 64 cells tops out near 3.9 TB/s), then every wavefront runs the lane-per-cell core
// on 4 of the tile's 16 strips, writing results back into the tile in place, and finally
// all threads store the tile.
//
// Every global store (and almost every load) is a whole 128-byte line: rows are laid out so
// that the first INTERIOR cell of a row is 128-byte aligned (the array base is shifted by
// LEAD = 16-mbc doubles, see pclaw.hip), and tile edges in the contiguous direction fall on
// multiples of 16 cells from there.  Unaligned 480-byte row pieces cost ~35 % of the
// achievable bandwidth on this part.
//   x pass (IXY=1): along = i (contiguous).  Tile = 4 rows x 244 cells (4 strips of 60 + halo):
//       240 = 15 x 16 updated cells per row.  LDS tile[m][row][244].  Wavefront w owns row w
//       and walks its 4 strips left to right.
//   y pass (IXY=2): along = j, across = i (contiguous).  Tile = 64 rows x 16 columns, one strip
//       per column.  LDS tile[m][row][16], column XOR-swizzled by the row (conflict-free column reads, no padding).  Wavefront w
//       owns columns w, w+4, w+8, w+12.
// step2ds semantics: the x pass sweeps every row (ghost rows too) and copies ghost columns
// through; the y pass sweeps every column and copies ghost rows through.
template <int IXY> struct TileShape {
    static constexpr int NSTRIP = IXY == 1 ? 4 : 1;                 // strips along the sweep per tile
    static constexpr int ALONG = NSTRIP * STRIP + 2 * HALO;         // cells loaded along the sweep
    static constexpr int ACROSS = IXY == 1 ? 4 : 16;                // cells across
#ifndef PCL_YTILE_PAD       /* 1: the round-1 layout (17th padding double per row) -- A/B builds for the bank-conflict counters */
#define PCL_YTILE_PAD 0
#endif
    static constexpr int PLANE = (IXY == 2 && PCL_YTILE_PAD) ? (ACROSS + 1) * ALONG : ACROSS * ALONG;
    static constexpr int UNITS = NSTRIP * ACROSS / 4;               // strips per wavefront
    __device__ static __forceinline__ int at(int m, int al, int ac) {
        // y pass: rows of 16 doubles, the column XOR-swizzled with bits 1..4 of the row instead of a 17th padding
        // double: a wavefront reading one column of 64 rows still hits 32 different 8-byte banks per half, and the five
        // Euler planes take 40960 B instead of 43520 -- FOUR workgroups per CU (160 KB) instead of three
#if PCL_YTILE_PAD
        return IXY == 1 ? (m * ACROSS + ac) * ALONG + al : m * PLANE + al * (ACROSS + 1) + ac;
#else
        return IXY == 1 ? (m * ACROSS + ac) * ALONG + al : (m * ALONG + al) * ACROSS + (ac ^ ((al >> 1) & (ACROSS - 1)));
#endif
    }
};
constexpr int LINE = 16;  // doubles per 128-byte line

// Boundary condition of one side as an index remap (solver.py:384-452): ghost index k of a dimension
// with n cells (ghosts included) reads interior index `src`; `neg` = reflecting (negate the normal
// momentum component), `cst` = constant inflow state.
struct VbcMap { int src; bool neg, cst; int side; };
__device__ __forceinline__ VbcMap vbc_map(int k, int n, int mbc, int lo, int hi) {
    VbcMap r{k, false, false, 0};
    if (k < mbc && lo >= 0) {
        r.side = 0;
        if (lo == 1) r.src = mbc;                       // outflow
        else if (lo == 2) r.src = n - 2 * mbc + k;      // periodic
        else if (lo == 3) { r.src = 2 * mbc - 1 - k; r.neg = true; }
        else r.cst = true;
    } else if (k >= n - mbc && hi >= 0) {
        r.side = 1;
        if (hi == 1) r.src = n - mbc - 1;
        else if (hi == 2) r.src = k - (n - 2 * mbc);    // q[n-mbc+t] = q[mbc+t]
        else if (hi == 3) { r.src = 2 * (n - mbc) - 1 - k; r.neg = true; }
        else r.cst = true;
    }
    return r;
}

// Workgroups are handed to the 8 XCDs round-robin (blockIdx % 8) and every XCD has its own L2.  Tiles that share
// cache lines (the unsplit kernels' 60-cell / 14-column pieces are not line-aligned) should therefore run on
// the SAME XCD, close in time, so that the shared lines are fetched once and the two partial-line stores merge
// in that L2 before they go to HBM.  This maps blockIdx to a logical index such that each XCD walks a
// contiguous range of logical indices (a bijection for any grid size).  PCL_TUNE_XCD=0 switches it off.
__device__ __forceinline__ int xcd_logical_block(int on) {
    if (!on) return blockIdx.x;
    const int nb = gridDim.x, x = blockIdx.x & 7, k = blockIdx.x >> 3;
    const int base = nb >> 3, rem = nb & 7;
    return x * base + (x < rem ? x : rem) + k;
}

// SRC: the instantiation that applies the fused source term while storing (y pass of the Euler solver only; the
// plain instantiation keeps its store loops untouched: a run-time switch there cost the memory-bound pass 4-5 %)
#ifndef PCL_SWEEP_OCC      /* workgroups per CU the aux-free instantiations are register-allocated for (A/B: 1 = no target) */
#define PCL_SWEEP_OCC 4
#endif
template <class RP, int IXY, bool CAPA, bool FWAVE, bool DIM1, bool SRC = false>
__global__ __launch_bounds__(256, (RP::NAUX == 0 && !CAPA) ? PCL_SWEEP_OCC : 1) void sweep_kernel(SweepArgs a, int ntiles_across, int ntiles_along) {
    using T = TileShape<IXY>;
    constexpr int MEQN = RP::MEQN;
    // tile planes: q(0..MEQN-1), the capacity function (CAPA), then the first RP::NAUX aux components for
    // Riemann solvers with cell-wise coefficients (variable-coefficient problems)
    constexpr int NAUX = RP::NAUX, PAUX = MEQN + (CAPA ? 1 : 0);
    constexpr int NP = PAUX + NAUX;
    __shared__ __attribute__((aligned(16))) double tile[NP * T::PLANE];

    // extents of the swept (along) and transverse (across) directions, ghost cells included
    const int n_along = IXY == 1 ? a.I : a.J;
    const int n_across = IXY == 1 ? a.J : a.I;
    const int m_along = IXY == 1 ? a.mx : a.my;  // interior cells along the sweep
    // Block order: the memory-contiguous direction varies fastest, so the workgroups in flight at
    // any moment stream whole rows.  (With the across index fastest in the x pass, all resident
    // workgroups read the same 2 KB column band of ~1000 rows whose pitch is 32 KB + 128 B: the
    // requests pile up on a few HBM channels and the pass drops to ~3.9 TB/s even as a pure copy.)
    // (the XCD-aware order of xcd_logical_block was tried here too: x pass unchanged, y pass +8 % -- these tiles
    // are line-aligned and share almost nothing)
    int tb = IXY == 1 ? blockIdx.x / ntiles_along : blockIdx.x % ntiles_across;
    int ta = IXY == 1 ? blockIdx.x % ntiles_along : blockIdx.x / ntiles_across;
    if (IXY == 1 && !DIM1 && a.sub != 0) {  // workgroup-uniform: interior box / its complement
        int idx = blockIdx.x;
        const int bw = a.box[3] - a.box[2];
        if (a.sub == 1) {
            tb = a.box[0] + idx / bw;
            ta = a.box[2] + idx % bw;
        } else {
            const int top = a.box[0] * ntiles_along, bot = (ntiles_across - a.box[1]) * ntiles_along;
            if (idx < top) {
                tb = idx / ntiles_along;
                ta = idx % ntiles_along;
            } else if (idx < top + bot) {
                idx -= top;
                tb = a.box[1] + idx / ntiles_along;
                ta = idx % ntiles_along;
            } else {
                idx -= top + bot;
                const int side = ntiles_along - bw;
                const int k = idx % side;
                tb = a.box[0] + idx / side;
                ta = k < a.box[2] ? k : a.box[3] + (k - a.box[2]);
            }
        }
    }
    // y pass: column tiles start LEAD cells before cell 0 so that they sit on 128-byte lines
    const int b0 = IXY == 1 ? tb * T::ACROSS : tb * T::ACROSS - (LINE - a.mbc);
    const int a0 = a.mbc - HALO + ta * (T::NSTRIP * STRIP);

    // ---- cooperative load ------------------------------------------------------------------
    // Tiles that lie wholly inside the array (all but the last of a row / the last row band) move 16 bytes per lane:
    // global_load_dwordx4 / global_store_dwordx4, half as many memory instructions and address computations as the
    // 8-byte form.  Rows start 16-byte aligned (LEAD = 16 - mbc doubles behind a 128-byte aligned base, even tile
    // offsets), so every pair is aligned; x tiles keep the pair together in LDS (ds_write_b128), y tiles split it
    // (their LDS rows have the odd pitch 17).
    const bool full_tile = a0 + T::ALONG <= n_along && b0 >= 0 && b0 + T::ACROSS <= n_across && (a.mbc & 1) == 0;
    // a tile takes the remap path only where it touches a side whose boundary condition is evaluated here: next to a
    // neighbour block (vbc < 0) the ghost cells are real memory (the halo exchange filled them) and the tile loads
    // like an interior one -- the rim tiles of a decomposed block keep the 16-byte loads
    const bool vbc_tile = IXY == 1 && a.vbc_on &&
                          (DIM1 || (a0 < a.mbc && a.vbc[0] >= 0) || (a0 + T::ALONG > n_along - a.mbc && a.vbc[1] >= 0) ||
                           (b0 < a.mbc && a.vbc[2] >= 0) || (b0 + T::ACROSS > n_across - a.mbc && a.vbc[3] >= 0));
    if (IXY == 1 && full_tile && !vbc_tile) {
        constexpr int PAIRS = T::ALONG / 2, SLOTS = PAIRS * T::ACROSS;      // 122 pairs x 4 rows
#pragma unroll
        for (int k = 0; k < (SLOTS + 255) / 256; k++) {
            const int slot = threadIdx.x + 256 * k;
            if (slot < SLOTS) {
                const int ac = slot / PAIRS, al = 2 * (slot % PAIRS);
                const long g = (long)(b0 + ac) * a.pitch + (a0 + al);
#pragma unroll
                for (int m = 0; m < MEQN; m++)
                    *reinterpret_cast<double2 *>(&tile[T::at(m, al, ac)]) = ld_stream2(&a.qin[m * a.plane + g]);
                if constexpr (CAPA)
                    *reinterpret_cast<double2 *>(&tile[T::at(MEQN, al, ac)]) =
                        *reinterpret_cast<const double2 *>(&a.aux[(long)(a.mcapa - 1) * a.plane + g]);
#pragma unroll
                for (int m = 0; m < NAUX; m++)
                    *reinterpret_cast<double2 *>(&tile[T::at(PAUX + m, al, ac)]) =
                        *reinterpret_cast<const double2 *>(&a.aux[aux_idx<RP, IXY>(m) * a.plane + g]);
            }
        }
    } else if (IXY == 1) {
        const int al = threadIdx.x;
        if (al < T::ALONG) {
            int ga = a0 + al;
            ga = ga < n_along ? ga : n_along - 1;  // clamp: cells past the edge repeat the last one
            // only tiles that touch the ghost frame take the remap path (workgroup-uniform): the plain
            // path keeps its 20 loads per thread independent and in flight together
            const bool vb = vbc_tile;
            if (!vb) {  // one basic block: all 20 loads issue before the first LDS write waits
#pragma unroll
                for (int ac = 0; ac < T::ACROSS; ac++) {
                    int gb = b0 + ac;
                    gb = gb < n_across ? gb : n_across - 1;
                    const long g = (long)gb * a.pitch + ga;
#pragma unroll
                    for (int m = 0; m < MEQN; m++) tile[T::at(m, al, ac)] = a.qin[m * a.plane + g];
                    if constexpr (CAPA) tile[T::at(MEQN, al, ac)] = a.aux[(long)(a.mcapa - 1) * a.plane + g];
#pragma unroll
                    for (int m = 0; m < NAUX; m++) tile[T::at(PAUX + m, al, ac)] = a.aux[aux_idx<RP, IXY>(m) * a.plane + g];
                }
            } else {
                // qbc = Y(X(q)): x sides first, then y sides over the x-filled array (solver.py:354-381)
                const VbcMap mi = vbc_map(ga, n_along, a.mbc, a.vbc[0], a.vbc[1]);
#pragma unroll
                for (int ac = 0; ac < T::ACROSS; ac++) {
                    int gb = b0 + ac;
                    gb = gb < n_across ? gb : n_across - 1;
                    const VbcMap mj = DIM1 ? VbcMap{gb, false, false, 0}
                                           : vbc_map(gb, n_across, a.mbc, a.vbc[2], a.vbc[3]);
                    const long gs = (long)mj.src * a.pitch + mi.src;
                    // branch-free: every lane loads (its own cell or the BC's source cell), then selects
#pragma unroll
                    for (int m = 0; m < MEQN; m++) {
                        double v = a.qin[m * a.plane + gs];
                        if (m == 1) v = mi.neg ? -v : v;
                        const double cx = mi.side ? a.vconst[1][m] : a.vconst[0][m];
                        v = mi.cst ? cx : v;
                        if (m == 2) v = mj.neg ? -v : v;
                        const double cy = mj.side ? a.vconst[3][m] : a.vconst[2][m];
                        v = mj.cst ? cy : v;
                        tile[T::at(m, al, ac)] = v;
                    }
                    if constexpr (CAPA)
                        tile[T::at(MEQN, al, ac)] = a.aux[(long)(a.mcapa - 1) * a.plane + (long)gb * a.pitch + ga];
#pragma unroll
                    for (int m = 0; m < NAUX; m++)   // aux ghost cells are real memory (auxbc is filled once at setup)
                        tile[T::at(PAUX + m, al, ac)] = a.aux[aux_idx<RP, IXY>(m) * a.plane + (long)gb * a.pitch + ga];
                }
            }
        }
    } else if (full_tile) {
        const int pr = threadIdx.x % (T::ACROSS / 2), ac = 2 * pr;            // 8 pairs per 128-byte row segment
#pragma unroll
        for (int k = 0; k < T::ALONG; k += 256 / (T::ACROSS / 2)) {
            const int al = k + threadIdx.x / (T::ACROSS / 2);
            const long g = (long)(a0 + al) * a.pitch + (b0 + ac);
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                const double2 v = ld_stream2(&a.qin[m * a.plane + g]);
                tile[T::at(m, al, ac)] = v.x;
                tile[T::at(m, al, ac + 1)] = v.y;
            }
            if constexpr (CAPA) {
                const double2 v = *reinterpret_cast<const double2 *>(&a.aux[(long)(a.mcapa - 1) * a.plane + g]);
                tile[T::at(MEQN, al, ac)] = v.x;
                tile[T::at(MEQN, al, ac + 1)] = v.y;
            }
#pragma unroll
            for (int m = 0; m < NAUX; m++) {
                const double2 v = *reinterpret_cast<const double2 *>(&a.aux[aux_idx<RP, IXY>(m) * a.plane + g]);
                tile[T::at(PAUX + m, al, ac)] = v.x;
                tile[T::at(PAUX + m, al, ac + 1)] = v.y;
            }
        }
    } else {
        const int ac = threadIdx.x % T::ACROSS;
        int gb = b0 + ac;
        gb = gb < 0 ? 0 : (gb < n_across ? gb : n_across - 1);
#pragma unroll
        for (int k = 0; k < T::ALONG; k += 256 / T::ACROSS) {
            const int al = k + threadIdx.x / T::ACROSS;
            int ga = a0 + al;
            ga = ga < n_along ? ga : n_along - 1;
            const long g = (long)ga * a.pitch + gb;
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[T::at(m, al, ac)] = a.qin[m * a.plane + g];
            if constexpr (CAPA) tile[T::at(MEQN, al, ac)] = a.aux[(long)(a.mcapa - 1) * a.plane + g];
#pragma unroll
            for (int m = 0; m < NAUX; m++) tile[T::at(PAUX + m, al, ac)] = a.aux[aux_idx<RP, IXY>(m) * a.plane + g];
        }
    }
    __syncthreads();

    // ---- arithmetic: each wavefront walks its UNITS strips ---------------------------------------
    // The strips of one row (x pass) overlap by 2*HALO cells inside the tile and results go back
    // in place, so a strip's cells are read into registers BEFORE the previous strip's results
    // are written (one-deep software pipeline; the same wavefront owns the whole row).
    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    auto unit_ac = [&](int u) { return IXY == 1 ? wv : wv + 4 * u; };
    auto unit_al = [&](int u) { return (IXY == 1 ? u * STRIP : 0) + lane; };
    auto unit_live = [&](int u) {  // wave-uniform
        const int gb = b0 + unit_ac(u);
        // step2ds sweeps every transverse index, ghost rows too (step2ds.f:83-88)
        const bool across_ok = gb >= 0 && gb < n_across;
        const bool along_ok = a0 + (IXY == 1 ? u * STRIP : 0) + HALO < a.mbc + m_along;  // not all past the interior
        return u < T::UNITS && across_ok && along_ok;
    };
    double cflmax = 0.0;
    double qnext[MEQN], capanext = 1.0;
    bool have_next = false;  // wave-uniform: qnext holds unit u's cells
#pragma unroll
    for (int u = 0; u < T::UNITS; u++) {
        if (!unit_live(u)) { have_next = false; continue; }  // wave-uniform
        const int al = unit_al(u), ac = unit_ac(u);
        const int ca = a0 + al;  // this lane's cell index along the sweep
        const bool owned = (ca >= a.mbc) && (ca < a.mbc + m_along) && lane >= HALO && lane < WAVE - HALO;
        const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + m_along) && lane >= 1;
        double q[MEQN], qn[MEQN], capa = 1.0, dtdx_c = a.dtd;
        if (have_next) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) q[m] = qnext[m];
            capa = capanext;
        } else {
#pragma unroll
            for (int m = 0; m < MEQN; m++) q[m] = tile[T::at(m, al, ac)];
            if constexpr (CAPA) capa = tile[T::at(MEQN, al, ac)];
        }
        have_next = unit_live(u + 1);
        if (have_next) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) qnext[m] = tile[T::at(m, unit_al(u + 1), unit_ac(u + 1))];
            if constexpr (CAPA) capanext = tile[T::at(MEQN, unit_al(u + 1), unit_ac(u + 1))];
        }
        if constexpr (CAPA) dtdx_c = DIM1 ? a.dt / (a.dx * capa) : a.dtd / capa;
        double auxv[NAUX > 0 ? NAUX : 1];
#pragma unroll
        for (int m = 0; m < NAUX; m++) auxv[m] = tile[T::at(PAUX + m, al, ac)];   // aux planes are never overwritten
        if (a.ablate & 1) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) qn[m] = q[m];
        } else
            lane_core<RP, IXY, CAPA, FWAVE, DIM1>(q, dtdx_c, capa, cfl_ok, a, qn, cflmax, nullptr, nullptr, nullptr, auxv);
        if (owned) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[T::at(m, al, ac)] = qn[m];
        }
    }
    __syncthreads();

    // ---- cooperative store: owned interior cells + ghost cells along the sweep (copied through) ----
    // the source term fused into the last pass of a dim-split step (SweepArgs.src_id): interior cells only
    static_assert(!SRC || (IXY == 2 && !DIM1 && MEQN == 5 && RP::NAUX == 0), "fused source: y pass of the Euler solver");
    auto interior = [&](int ga, int gb) {
        return ga >= a.mbc && ga < a.mbc + m_along && gb >= a.mbc && gb < a.mbc + (IXY == 1 ? a.my : a.mx);
    };
    auto put = [&](int al, int ac) {
        const int ga = a0 + al, gb = b0 + ac;
        if (ga < n_along && gb >= 0 && gb < n_across) {
            const long g = IXY == 1 ? (long)gb * a.pitch + ga : (long)ga * a.pitch + gb;
            if constexpr (SRC) {
                double v[MEQN];
#pragma unroll
                for (int m = 0; m < MEQN; m++) v[m] = tile[T::at(m, al, ac)];
                if (interior(ga, gb)) euler_radial_source(v[0], v[1], v[2], v[3], a.aux[g], a.dt, a.src_p[0], a.src_p[1]);
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = v[m];
            } else {
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = tile[T::at(m, al, ac)];
            }
        }
    };
    if (IXY == 1 && full_tile) {
        constexpr int PAIRS = T::NSTRIP * STRIP / 2, SLOTS = PAIRS * T::ACROSS;   // cells a0+2 .. a0+241: 120 pairs x 4 rows
#pragma unroll
        for (int k = 0; k < (SLOTS + 255) / 256; k++) {
            const int slot = threadIdx.x + 256 * k;
            if (slot < SLOTS) {
                const int ac = slot / PAIRS, al = HALO + 2 * (slot % PAIRS);
                const long g = (long)(b0 + ac) * a.pitch + (a0 + al);
#pragma unroll
                for (int m = 0; m < MEQN; m++)
                    st_stream2(&a.qout[m * a.plane + g], *reinterpret_cast<const double2 *>(&tile[T::at(m, al, ac)]));
            }
        }
        const int t = threadIdx.x;
        if (t < 2 * HALO) {          // the tile's own halo cells, only where they are ghost cells
            const int al = t < HALO ? t : T::ALONG - 2 * HALO + t;
            const int ga = a0 + al;
            if (ga < a.mbc || ga >= a.mbc + m_along)
                for (int ac = 0; ac < T::ACROSS; ac++) put(al, ac);
        }
    } else if (IXY == 1) {
        const int t = threadIdx.x;
        if (t < T::NSTRIP * STRIP) {  // cells a0+2 .. a0+241: 15 whole lines per row
#pragma unroll
            for (int ac = 0; ac < T::ACROSS; ac++) put(HALO + t, ac);
        } else if (t < T::NSTRIP * STRIP + 2 * HALO) {  // the tile's own halo cells, only where they are ghosts
            const int k = t - T::NSTRIP * STRIP;
            const int al = k < HALO ? k : T::ALONG - 2 * HALO + k;
            const int ga = a0 + al;
            if (ga < a.mbc || ga >= a.mbc + m_along)
                for (int ac = 0; ac < T::ACROSS; ac++) put(al, ac);
        }
    } else if (full_tile) {
        const int pr = threadIdx.x % (T::ACROSS / 2), ac = 2 * pr;
#pragma unroll
        for (int k = 0; k < T::ALONG; k += 256 / (T::ACROSS / 2)) {
            const int al = k + threadIdx.x / (T::ACROSS / 2);
            const int ga = a0 + al;
            const bool inner = (ga >= a.mbc) && (ga < a.mbc + m_along);
            if (inner ? (al >= HALO && al < T::ALONG - HALO) : true) {
                const long g = (long)ga * a.pitch + (b0 + ac);
                if constexpr (SRC) {
                    double vx[MEQN], vy[MEQN];
#pragma unroll
                    for (int m = 0; m < MEQN; m++) {
                        vx[m] = tile[T::at(m, al, ac)];
                        vy[m] = tile[T::at(m, al, ac + 1)];
                    }
                    if (interior(ga, b0 + ac)) euler_radial_source(vx[0], vx[1], vx[2], vx[3], a.aux[g], a.dt, a.src_p[0], a.src_p[1]);
                    if (interior(ga, b0 + ac + 1)) euler_radial_source(vy[0], vy[1], vy[2], vy[3], a.aux[g + 1], a.dt, a.src_p[0], a.src_p[1]);
#pragma unroll
                    for (int m = 0; m < MEQN; m++) {
                        double2 v;
                        v.x = vx[m];
                        v.y = vy[m];
                        st_stream2(&a.qout[m * a.plane + g], v);
                    }
                } else {
#pragma unroll
                    for (int m = 0; m < MEQN; m++) {
                        double2 v;
                        v.x = tile[T::at(m, al, ac)];
                        v.y = tile[T::at(m, al, ac + 1)];
                        st_stream2(&a.qout[m * a.plane + g], v);
                    }
                }
            }
        }
    } else {
        const int ac = threadIdx.x % T::ACROSS;
#pragma unroll
        for (int k = 0; k < T::ALONG; k += 256 / T::ACROSS) {
            const int al = k + threadIdx.x / T::ACROSS;
            const int ga = a0 + al;
            const bool inner = (ga >= a.mbc) && (ga < a.mbc + m_along);
            if (inner ? (al >= HALO && al < T::ALONG - HALO) : true) put(al, ac);
        }
    }
    cfl_publish(a.cfl, cfl_value<CAPA>(cflmax, a.dtd));
}

// ---- unsplit algorithm (step2.f / step2qcor.f), with and without a capacity function -------------------
// A workgroup of U_WAVES wavefronts takes U_WAVES consecutive slices (x phase: rows of one 64-cell
// strip; y phase: columns of one 64-row strip).  Every wavefront computes its slice's pieces with the
// TRANS core, publishes dt/d*gadd(.,1,.) and dt/d*gadd(.,2,.) in LDS, and after one barrier each of
// the U_WAVES-2 inner slices adds its neighbours' transverse contributions in the reference's order
// (step2.f:130-137 / :214-218).  Tiles overlap by two slices (U_WAVES/(U_WAVES-2) recompute) instead
// of moving 8 scratch planes through HBM.

template <class RP, bool FWAVE, int U_WAVES, bool CAPA = false>
__global__ __launch_bounds__(U_WAVES *WAVE) void unsplit_x_kernel(SweepArgs a, int nstrips) {
    constexpr int MEQN = RP::MEQN;
    constexpr int U_OUT = U_WAVES - 2;
    __shared__ double gm[U_WAVES][MEQN][WAVE], gp[U_WAVES][MEQN][WAVE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = threadIdx.x / WAVE;
    const int bid = xcd_logical_block(a.xcd);
    const int ta = bid % nstrips, tr = bid / nstrips;
    if (a.sub != 0) {   // decomposed run: the tiles that read no ghost cell (box) in one launch, the rim in another
        const bool inside = tr >= a.box[0] && tr < a.box[1] && ta >= a.box[2] && ta < a.box[3];
        if ((a.sub == 1) != inside) return;             // workgroup-uniform, before any barrier
    }
    const int a0 = a.mbc - HALO + ta * STRIP;
    const int row = a.mbc - 1 + tr * U_OUT + w;        // slices j = 0 .. my+1  <->  rows mbc-1 .. mbc+my
    const bool slice_ok = row <= a.mbc + a.my;           // wave-uniform
    const int ca = a0 + lane;
    const int cc = ca < a.I ? ca : a.I - 1;
    const int rc = row < a.J ? row : a.J - 1;
    const long g = (long)rc * a.pitch + cc;
    double q[MEQN], qadd[MEQN], df[MEQN], g1[MEQN], g2[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) q[m] = ld_stream(&a.qin[m * a.plane + g]);
    double cflmax = 0.0;
    const bool cfl_ok = slice_ok && (ca >= a.mbc) && (ca <= a.mbc + a.mx) && lane >= 1;
    // solver aux of this row and of the rows below / above it (step2.f:97-101: aux1, aux2, aux3)
    constexpr int NAUX = RP::NAUX;
    constexpr int NT = naux_t<RP>() > 0 ? naux_t<RP>() : 1;
    double auxv[NAUX > 0 ? NAUX : 1], auxo[NT], auxb[NT], auxa[NT];
    if constexpr (NAUX > 0) {
        const int rb = rc > 0 ? rc - 1 : 0, ra = rc + 1 < a.J ? rc + 1 : a.J - 1;
#pragma unroll
        for (int m = 0; m < NAUX; m++) auxv[m] = a.aux[aux_idx<RP, 1>(m) * a.plane + g];
#pragma unroll
        for (int m = 0; m < NT; m++) {
            const long pl = auxt_idx<RP, 1>(m) * a.plane;
            auxo[m] = a.aux[pl + g];
            auxb[m] = a.aux[pl + (long)rb * a.pitch + cc];
            auxa[m] = a.aux[pl + (long)ra * a.pitch + cc];
        }
    }
    // capacity function (step2.f:86-90): dtdx1d = dtdx/capa; every increment is divided by the TARGET cell's capa
    double capa = 1.0, dtdx_c = a.dtd;
    if constexpr (CAPA) { capa = a.aux[(long)(a.mcapa - 1) * a.plane + g]; dtdx_c = a.dtd / capa; }
    lane_core<RP, 1, CAPA, FWAVE, false, true>(q, dtdx_c, capa, cfl_ok, a, qadd, cflmax, df, g1, g2, auxv, auxb, auxa,
                                               auxo);
    // conservation fix on the sphere (step2qcor.f:146-159 + qcor.f): this cell's edge and the next cell's
    double qc[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) qc[m] = 0.0;
    if constexpr (has_qcor<RP>() && CAPA) {
        double er_[6], qc4[4];
#pragma unroll
        for (int k = 0; k < 6; k++) er_[k] = from_right(auxv[k]);
        RP::template qcor<1>(q, auxv, er_, auxv + 6, a.par, qc4);
#pragma unroll
        for (int m = 0; m < MEQN; m++) qc[m] = qc4[m];
    }
#pragma unroll
    for (int m = 0; m < MEQN; m++) {
        gm[w][m][lane] = a.dtd_t * g1[m];
        gp[w][m][lane] = a.dtd_t * g2[m];
    }
    __syncthreads();
    const bool out_row = w >= 1 && w <= U_OUT && row >= a.mbc && row < a.mbc + a.my;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.mx) && lane >= HALO && lane < WAVE - HALO;
    if (out_row && owned) {
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            double v;
            if constexpr (CAPA) {    // step2.f:145-152
                v = q[m] + gp[w - 1][m][lane] / capa;
                v = v + qadd[m] - (a.dtd * df[m] + a.dtd_t * (g2[m] - g1[m])) / capa;
                if constexpr (has_qcor<RP>()) v = v - a.dtd * qc[m] / capa;       // step2qcor.f:157
                v = v - gm[w + 1][m][lane] / capa;
            } else {
                v = q[m] + gp[w - 1][m][lane];                                      // from slice j-1
                v = v + qadd[m] - a.dtd * df[m] - a.dtd_t * (g2[m] - g1[m]);       // slice j
                v = v - gm[w + 1][m][lane];                                         // from slice j+1
            }
            st_stream(&a.qout[m * a.plane + g], v);
        }
    }
    cfl_publish(a.cfl, cfl_value<CAPA>(cflmax, a.dtd));
}

// y phase: qx = result of the x phase (read and overwritten cell by cell), a.qin = qold
// SRC: the fused Godunov-split source term, as in sweep_kernel (own instantiation, Euler solver without capa)
template <class RP, bool FWAVE, int U_WAVES, bool CAPA = false, bool SRC = false>
__global__ __launch_bounds__(U_WAVES *WAVE) void unsplit_y_kernel(SweepArgs a, int ntiles_i, const double *qx) {
    constexpr int MEQN = RP::MEQN;
    constexpr int U_OUT = U_WAVES - 2;
    constexpr int TP = U_WAVES + 1;
    __shared__ double tile[MEQN][WAVE][TP];
    __shared__ double gm[U_WAVES][MEQN][WAVE], gp[U_WAVES][MEQN][WAVE];
    // aux planes of the solver, columns i0-1 .. i0+U_WAVES (the transverse solves read the neighbouring columns):
    // loaded with coalesced row segments like q; per-lane loads of a column are `pitch` apart (64 lines per
    // instruction, 9 + 27 of them for the sphere solver)
    // (where all of it fits the 160 KB: the sphere solver's 16 planes do with 8 slices per workgroup, its default)
    constexpr int AC = U_WAVES + 2, AP = AC + 1;
    constexpr long LDS_REST = (long)sizeof(double) * (MEQN * WAVE * TP + 2 * U_WAVES * MEQN * WAVE);
    constexpr int NPL = LDS_REST + (long)sizeof(double) * aux_planes_y<RP>() * WAVE * AP <= 160 * 1024 ? aux_planes_y<RP>() : 0;
    __shared__ double atile[NPL > 0 ? NPL : 1][NPL > 0 ? WAVE : 1][NPL > 0 ? AP : 1];
    const int bid = xcd_logical_block(a.xcd);
    const int ti = bid % ntiles_i, tj = bid / ntiles_i;
    const int i0 = a.mbc - 1 + ti * U_OUT;               // slices i = 0 .. mx+1  <->  columns mbc-1 .. mbc+mx
    const int j0 = a.mbc - HALO + tj * STRIP;
    {   // cooperative load of qold: 16 lanes per row segment
        const int c = threadIdx.x % U_WAVES, r = threadIdx.x / U_WAVES;
        int gi = i0 + c, gj = j0 + r;
        gi = gi < a.I ? gi : a.I - 1;
        gj = gj < a.J ? gj : a.J - 1;
        const long g = (long)gj * a.pitch + gi;
#pragma unroll
        for (int m = 0; m < MEQN; m++) tile[m][r][c] = a.qin[m * a.plane + g];
    }
    if constexpr (NPL > 0) {
        for (int idx = threadIdx.x; idx < WAVE * AC; idx += U_WAVES * WAVE) {
            const int c = idx % AC, r = idx / AC;
            int gi = i0 - 1 + c, gj = j0 + r;
            gi = gi < 0 ? 0 : (gi < a.I ? gi : a.I - 1);
            gj = gj < a.J ? gj : a.J - 1;
            const long g = (long)gj * a.pitch + gi;
#pragma unroll
            for (int p = 0; p < NPL; p++)
                if (aux_plane_used_y<RP>(p)) atile[p][r][c] = a.aux[p * a.plane + g];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & (WAVE - 1);
    const int w = threadIdx.x / WAVE;
    const int col = i0 + w;
    const bool slice_ok = col <= a.mbc + a.mx;
    const int cj = j0 + lane;
    double q[MEQN], qadd[MEQN], df[MEQN], g1[MEQN], g2[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) q[m] = tile[m][lane][w];
    double cflmax = 0.0;
    const bool cfl_ok = slice_ok && (cj >= a.mbc) && (cj <= a.mbc + a.my) && lane >= 1;
    // solver aux of this column and of the columns to its left / right (step2.f:177-181)
    constexpr int NAUX = RP::NAUX;
    constexpr int NT = naux_t<RP>() > 0 ? naux_t<RP>() : 1;
    double auxv[NAUX > 0 ? NAUX : 1], auxo[NT], auxb[NT], auxa[NT];
    if constexpr (NAUX > 0 && NPL > 0) {     // column w of the tile = column w+1 of atile
#pragma unroll
        for (int m = 0; m < NAUX; m++) auxv[m] = atile[aux_idx<RP, 2>(m)][lane][w + 1];
#pragma unroll
        for (int m = 0; m < NT; m++) {
            auxo[m] = atile[auxt_idx<RP, 2>(m)][lane][w + 1];
            auxb[m] = atile[auxt_idx<RP, 2>(m)][lane][w];
            auxa[m] = atile[auxt_idx<RP, 2>(m)][lane][w + 2];
        }
    } else if constexpr (NAUX > 0) {
        const int gj = cj < a.J ? cj : a.J - 1;
        const int c0 = col < a.I ? col : a.I - 1;
        const int cb = c0 > 0 ? c0 - 1 : 0, cn = c0 + 1 < a.I ? c0 + 1 : a.I - 1;
#pragma unroll
        for (int m = 0; m < NAUX; m++) auxv[m] = a.aux[aux_idx<RP, 2>(m) * a.plane + (long)gj * a.pitch + c0];
#pragma unroll
        for (int m = 0; m < NT; m++) {
            const long pl = auxt_idx<RP, 2>(m) * a.plane;
            auxo[m] = a.aux[pl + (long)gj * a.pitch + c0];
            auxb[m] = a.aux[pl + (long)gj * a.pitch + cb];
            auxa[m] = a.aux[pl + (long)gj * a.pitch + cn];
        }
    }
    double capa = 1.0, dtdx_c = a.dtd;
    if constexpr (CAPA) {
        const int gj = cj < a.J ? cj : a.J - 1;
        const int c0 = col < a.I ? col : a.I - 1;
        capa = a.aux[(long)(a.mcapa - 1) * a.plane + (long)gj * a.pitch + c0];
        dtdx_c = a.dtd / capa;
        // every wavefront has its qold column in registers: the tile is free to take the x-phase result
        __syncthreads();
        const int c = threadIdx.x % U_WAVES, r = threadIdx.x / U_WAVES;
        int gi = i0 + c, gr = j0 + r;
        gi = gi < a.I ? gi : a.I - 1;
        gr = gr < a.J ? gr : a.J - 1;
#pragma unroll
        for (int m = 0; m < MEQN; m++) tile[m][r][c] = qx[m * a.plane + (long)gr * a.pitch + gi];
    }
    lane_core<RP, 2, CAPA, FWAVE, false, true>(q, dtdx_c, capa, cfl_ok, a, qadd, cflmax, df, g1, g2, auxv, auxb, auxa,
                                               auxo);
    double qc[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) qc[m] = 0.0;
    if constexpr (has_qcor<RP>() && CAPA) {   // step2qcor.f:232-245
        double er_[6], qc4[4];
#pragma unroll
        for (int k = 0; k < 6; k++) er_[k] = from_right(auxv[k]);
        RP::template qcor<2>(q, auxv, er_, auxv + 6, a.par, qc4);
#pragma unroll
        for (int m = 0; m < MEQN; m++) qc[m] = qc4[m];
    }
#pragma unroll
    for (int m = 0; m < MEQN; m++) {
        gm[w][m][lane] = a.dtd_t * g1[m];
        gp[w][m][lane] = a.dtd_t * g2[m];
        if constexpr (!CAPA)
            tile[m][lane][w] = qadd[m] - a.dtd * df[m] - a.dtd_t * (g2[m] - g1[m]);  // only this wave reads column w
    }
    __syncthreads();
    if constexpr (CAPA) {
        // step2.f:227-234: each increment divided by this cell's capa, in slice order i-1, i, i+1; the result goes
        // back into the tile for the coalesced store
        if (w >= 1 && w <= U_OUT) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                double v = tile[m][lane][w] + gp[w - 1][m][lane] / capa;
                v = v + qadd[m] - (a.dtd * df[m] + a.dtd_t * (g2[m] - g1[m])) / capa;
                if constexpr (has_qcor<RP>()) v = v - a.dtd * qc[m] / capa;
                v = v - gm[w + 1][m][lane] / capa;
                tile[m][lane][w] = v;
            }
        }
        __syncthreads();
        const int c = threadIdx.x % U_WAVES, r = threadIdx.x / U_WAVES;
        const int gi = i0 + c, gj = j0 + r;
        const bool ok = c >= 1 && c <= U_OUT && gi >= a.mbc && gi < a.mbc + a.mx && r >= HALO && r < WAVE - HALO &&
                        gj >= a.mbc && gj < a.mbc + a.my;
        if (ok) {
            const long g = (long)gj * a.pitch + gi;
#pragma unroll
            for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = tile[m][r][c];
        }
    } else {   // cooperative combine + store: q6 = ((q3 + gp'(i-1)) + mid(i)) - gm'(i+1)
        const int c = threadIdx.x % U_WAVES, r = threadIdx.x / U_WAVES;
        const int gi = i0 + c, gj = j0 + r;
        const bool ok = c >= 1 && c <= U_OUT && gi >= a.mbc && gi < a.mbc + a.mx && r >= HALO && r < WAVE - HALO &&
                        gj >= a.mbc && gj < a.mbc + a.my;
        if (ok) {
            const long g = (long)gj * a.pitch + gi;
            if constexpr (SRC) {
                static_assert(MEQN == 5 && !CAPA, "fused source: Euler solver without a capacity function");
                double w[MEQN];
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    double v = qx[m * a.plane + g] + gp[c - 1][m][r];
                    v = v + tile[m][r][c];
                    w[m] = v - gm[c + 1][m][r];
                }
                euler_radial_source(w[0], w[1], w[2], w[3], a.aux[g], a.dt, a.src_p[0], a.src_p[1]);
#pragma unroll
                for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = w[m];
            } else {
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    double v = qx[m * a.plane + g] + gp[c - 1][m][r];
                    v = v + tile[m][r][c];
                    v = v - gm[c + 1][m][r];
                    a.qout[m * a.plane + g] = v;
                }
            }
        }
    }
    cfl_publish(a.cfl, cfl_value<CAPA>(cflmax, a.dtd));
}

// ---- y phase, marching form: line-aligned columns, no recomputed slices -------------------------------------------
// unsplit_y_kernel's tiles give 14 finished columns for 16 computed ones, and 14 is not a multiple of the 16 doubles
// of a 128-byte line: every line of the result is completed by two workgroups (WRITE_SIZE 1.66x the planes written,
// profiles/r02_pmc_hbm.json).  Here a workgroup of 16 wavefronts walks along i through one band of 64 rows.  Step T
// computes the 16 slices (columns) mbc+16T+1 .. mbc+16T+16 -- ONE column past a line boundary -- and finishes the 16
// columns mbc+16T .. mbc+16T+15, exactly one line per row: the first of them is the last slice of the previous step
// (its own increment and the two neighbour contributions it was still missing are carried over in LDS), the other 15
// are this step's slices 0..14 (slice 15 waits for the next step's first slice).  Every x-phase value is read and
// every result is written as a whole aligned line, no slice is computed twice, and the march range is cut into
// segments (one warm-up step each, whose only purpose is the carry) so that the grid still has ~4 workgroups per CU.
// The combination per cell is the reference's: q6 = ((q3 + gp'(i-1)) + mid(i)) - gm'(i+1)   (step2.f:214-218).
// Solvers without aux arrays, no capacity function (the others keep unsplit_y_kernel).
// NWV wavefronts per workgroup, each takes 16 / NWV of the step's 16 slices in turn: 16 (128 VGPRs per lane: the Euler
// core spills ~29 of them) or 8 (256 VGPRs, no spills, two slices per wavefront and step).
template <class RP, bool FWAVE, bool SRC = false, int NWV = 16>
__global__ __launch_bounds__(NWV *WAVE) void unsplit_ym_kernel(SweepArgs a, int ntj, int seg, const double *qx) {
    // slots 2..17: this step's slices; the carry (slices 14, 15 of the previous step) sits in slots 0, 1 or 18, 19 in turn,
    // so that a step can write the next carry while its own is still being read
    constexpr int MEQN = RP::MEQN, UW = 16, NS = UW + 4, SPW = UW / NWV, NT = NWV * WAVE;
    static_assert(RP::NAUX == 0, "solvers without aux arrays");
    static_assert(UW % NWV == 0, "16 slices per step");
    __shared__ double tile[MEQN][WAVE][NS + 1];                    // qold columns in, then each slice's own increment "mid"
    __shared__ double gp[NS][MEQN][WAVE], gm[UW][MEQN][WAVE];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int tj = blockIdx.x % ntj, ts = blockIdx.x / ntj;
    const int nlines = (a.mx + UW - 1) / UW;
    const int L0 = ts * seg, L1 = L0 + seg < nlines ? L0 + seg : nlines;
    const int j0 = a.mbc - HALO + tj * STRIP;
    const int cj = j0 + lane;
    double cflmax = 0.0;
    for (int T = L0 - 1; T < L1; T++) {
        const int c0 = a.mbc + UW * T + 1;                         // first computed column; finished: c0-1 .. c0+14
        // cooperative load of qold: 16 lanes per row segment (one column past the line boundary: two lines per row,
        // the second one is the next step's first)
#pragma unroll
        for (int k = 0; k < UW * WAVE / NT; k++) {
            const int id = threadIdx.x + k * NT;
            const int c = id % UW, r = id / UW;
            int gi = c0 + c, gj = j0 + r;
            gi = gi < 0 ? 0 : (gi < a.I ? gi : a.I - 1);
            gj = gj < a.J ? gj : a.J - 1;
            const long g = (long)gj * a.pitch + gi;
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[m][r][c + 2] = a.qin[m * a.plane + g];
        }
        __syncthreads();
#pragma unroll 1
        for (int k = 0; k < SPW; k++) {
            const int sl = w + NWV * k;                             // this wavefront's slice of the step
            const int col = c0 + sl;
            const bool slice_ok = col >= a.mbc - 1 && col <= a.mbc + a.mx;      // slices 0 .. mx+1 (wave-uniform)
            if (slice_ok && !(T < L0 && sl < UW - 2)) {             // the warm-up step only needs its last two slices
                double q[MEQN], qadd[MEQN], df[MEQN], g1[MEQN], g2[MEQN];
#pragma unroll
                for (int m = 0; m < MEQN; m++) q[m] = tile[m][lane][sl + 2];
                const bool cfl_ok = (cj >= a.mbc) && (cj <= a.mbc + a.my) && lane >= 1;
                lane_core<RP, 2, false, FWAVE, false, true>(q, a.dtd, 1.0, cfl_ok, a, qadd, cflmax, df, g1, g2);
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    gm[sl][m][lane] = a.dtd_t * g1[m];
                    gp[sl + 2][m][lane] = a.dtd_t * g2[m];
                    tile[m][lane][sl + 2] = qadd[m] - a.dtd * df[m] - a.dtd_t * (g2[m] - g1[m]);   // only this wave reads the column
                }
            }
        }
        __syncthreads();
        const int cin = ((T - (L0 - 1)) & 1) ? UW + 2 : 0, cout = cin ? 0 : UW + 2;       // carry slots read / written by this step
        if (w >= NWV - 2 && T + 1 < L1) {   // carry: slices 14, 15 go to the next step (nobody reads `cout` during this step)
            const int sl = UW - 2 + (w - (NWV - 2));
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                gp[cout + sl - (UW - 2)][m][lane] = gp[sl + 2][m][lane];
                tile[m][lane][cout + sl - (UW - 2)] = tile[m][lane][sl + 2];
            }
        }
        if (T >= L0) {   // finish the line: column c0-1+c = the previous step's last slice (c = 0) or this step's slice c-1
#pragma unroll
            for (int k = 0; k < UW * WAVE / NT; k++) {
                const int id = threadIdx.x + k * NT;
                const int c = id % UW, r = id / UW;
                const int gi = c0 - 1 + c, gj = j0 + r;
                const bool ok = gi >= a.mbc && gi < a.mbc + a.mx && r >= HALO && r < WAVE - HALO && gj >= a.mbc && gj < a.mbc + a.my;
                if (ok) {
                    const long g = (long)gj * a.pitch + gi;
                    double v[MEQN];
#pragma unroll
                    for (int m = 0; m < MEQN; m++) {
                        double x = qx[m * a.plane + g] + gp[c < 2 ? cin + c : c][m][r];     // from slice i-1 = column c0-2+c
                        x = x + tile[m][r][c < 1 ? cin + 1 : c + 1];                        // the slice's own increment
                        v[m] = x - gm[c][m][r];                                             // from slice i+1 (this step's slice c)
                    }
                    if constexpr (SRC) {
                        static_assert(!SRC || MEQN == 5, "fused source: Euler solver");
                        euler_radial_source(v[0], v[1], v[2], v[3], a.aux[g], a.dt, a.src_p[0], a.src_p[1]);
                    }
#pragma unroll
                    for (int m = 0; m < MEQN; m++) a.qout[m * a.plane + g] = v[m];
                }
            }
        }
        __syncthreads();       // the next step's load overwrites slots 2..17
    }
    cfl_publish(a.cfl, cfl_value<false>(cflmax, a.dtd));
}

// ---- 3-D dimension-split sweep (step3ds.f:108-374 + flux3.f:168-258) ---------------------------------
// One kernel for the three directions: DIR = 1 (along i, memory-contiguous), 2 (along j), 3 (along k).
// The array is q[m][k][j][i]; a launch describes the sweep by strides: s_al (along the sweep), s_ac (the
// "across" index that shares a tile) and s_b (the batch index, blockIdx.y).  Tile = 64 cells along x 16
// across, one lane per cell like the 2-D kernel; along-contiguous tiles sit in LDS as [ac][al], the others
// as [al][ac] with pitch 17 (coalesced 128-byte row segments in, conflict-free column reads out).
// Slices are swept only for transverse indices 0..m+1 (one ghost layer, step3ds.f:110-111,176-177,245-246);
// every other cell is copied through, so the output array is complete.
// CAPA: capacity function aux(mcapa) (step3ds.f:138-141,196-200: dtdx1d = dtdx / capa per cell, the correction-flux
// difference of the update divided by the cell's capa); it rides in the tile as one more plane.
template <class RP, int DIR, bool CAPA = false>
__global__ __launch_bounds__(256) void sweep3_kernel(SweepArgs a, int ntiles_ac, int ntiles_al) {
    constexpr int MEQN = RP::MEQN, NAUX = RP::NAUX, NP = MEQN + NAUX + (CAPA ? 1 : 0);
    // DIR 1 (along i): 4 rows x 244 cells, each wavefront walks the 4 strips of its row: 240 updated cells =
    // 15 whole lines per row, like the 2-D x pass.  DIR 2, 3: 64 cells along x 16 columns, one strip per column.
    constexpr int NSTRIP = DIR == 1 ? 4 : 1;
    constexpr int ADV = NSTRIP * STRIP, ALONG = ADV + 2 * HALO;
    constexpr int AC = DIR == 1 ? 4 : 16;
    constexpr int PLANE = DIR == 1 ? AC * ALONG : ALONG * (AC + 1);
    constexpr int UNITS = NSTRIP * AC / 4;
    __shared__ double tile[NP * PLANE];
    auto at = [](int m, int al, int ac) { return m * PLANE + (DIR == 1 ? ac * ALONG + al : al * (AC + 1) + ac); };

    const int tb = DIR == 1 ? blockIdx.x / ntiles_al : blockIdx.x % ntiles_ac;
    const int ta = DIR == 1 ? blockIdx.x % ntiles_al : blockIdx.x / ntiles_ac;
    const int bt = blockIdx.y;
    // across = i (DIR 2, 3): column tiles start LEAD cells before cell 0 so that every 16-cell row segment
    // is one 128-byte line (the array base is shifted so that cell mbc starts a line, pclaw.hip)
    const int b0 = DIR == 1 ? tb * AC : tb * AC - (LINE - a.mbc);
    const int a0 = a.mbc - HALO + ta * ADV;
    const long base = (long)bt * a.s_b;

    // cooperative load: memory-contiguous index fastest across the threads
    auto load = [&](int al, int ac) {
        int ga = a0 + al, gb = b0 + ac;
        ga = ga < a.n_al ? ga : a.n_al - 1;
        gb = gb < 0 ? 0 : (gb < a.n_ac ? gb : a.n_ac - 1);
        const long g = base + (long)ga * a.s_al + (long)gb * a.s_ac;
#pragma unroll
        for (int m = 0; m < MEQN; m++) tile[at(m, al, ac)] = ld_stream(&a.qin[m * a.plane + g]);
#pragma unroll
        for (int m = 0; m < NAUX; m++) tile[at(MEQN + m, al, ac)] = a.aux[aux_idx<RP, DIR>(m) * a.plane + g];
        if constexpr (CAPA) tile[at(MEQN + NAUX, al, ac)] = a.aux[(long)(a.mcapa - 1) * a.plane + g];
    };
    if (DIR == 1) {
        if ((int)threadIdx.x < ALONG) {
#pragma unroll
            for (int ac = 0; ac < AC; ac++) load(threadIdx.x, ac);
        }
    } else {
#pragma unroll
        for (int k = 0; k < ALONG; k += 256 / AC) load(k + threadIdx.x / AC, threadIdx.x % AC);
    }
    __syncthreads();

    const int lane = threadIdx.x & (WAVE - 1), wv = threadIdx.x / WAVE;
    const bool batch_live = bt >= a.lo_b && bt <= a.hi_b;
    auto unit_ac = [&](int u) { return DIR == 1 ? wv : wv + 4 * u; };
    auto unit_al = [&](int u) { return (DIR == 1 ? u * STRIP : 0) + lane; };
    auto unit_live = [&](int u) {  // wave-uniform: the slice is swept and the strip reaches interior cells
        const int gb = b0 + unit_ac(u);
        return u < UNITS && batch_live && gb >= a.lo_ac && gb <= a.hi_ac &&
               a0 + (DIR == 1 ? u * STRIP : 0) + HALO < a.mbc + a.m_al;
    };
    double cflmax = 0.0;
    // strips of one row (DIR 1) overlap by 2*HALO cells in the tile and results go back in place: the next
    // strip's cells are read before this strip's results are written (LDS keeps a wavefront's order)
    double q[MEQN], auxv[NAUX > 0 ? NAUX : 1], capa = 1.0;
    bool have = false;
    auto fetch = [&](int u) {
#pragma unroll
        for (int m = 0; m < MEQN; m++) q[m] = tile[at(m, unit_al(u), unit_ac(u))];
#pragma unroll
        for (int m = 0; m < NAUX; m++) auxv[m] = tile[at(MEQN + m, unit_al(u), unit_ac(u))];
        if constexpr (CAPA) capa = tile[at(MEQN + NAUX, unit_al(u), unit_ac(u))];
    };
#pragma unroll
    for (int u = 0; u < UNITS; u++) {
        if (!unit_live(u)) { have = false; continue; }
        const int al = unit_al(u), ac = unit_ac(u);
        const int ca = a0 + al;
        const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.m_al) && lane >= HALO && lane < WAVE - HALO;
        const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.m_al) && lane >= 1;
        if (!have) fetch(u);
        double qn[MEQN];
        lane_core<RP, DIR, CAPA, false, false, false, true>(q, CAPA ? a.dtd / capa : a.dtd, capa, cfl_ok, a, qn, cflmax, nullptr,
                                                            nullptr, nullptr, auxv);
        have = unit_live(u + 1);
        if (have) fetch(u + 1);  // issued ahead of the writes below
        if (owned) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[at(m, al, ac)] = qn[m];
        }
    }
    __syncthreads();

    // cooperative store: cells this tile owns along the sweep + the ghost cells at the ends of the sweep
    auto put = [&](int al, int ac) {
        const int ga = a0 + al, gb = b0 + ac;
        if (ga < a.n_al && gb >= 0 && gb < a.n_ac) {
            const long g = base + (long)ga * a.s_al + (long)gb * a.s_ac;
#pragma unroll
            for (int m = 0; m < MEQN; m++) st_stream(&a.qout[m * a.plane + g], tile[at(m, al, ac)]);
        }
    };
    if (DIR == 1) {
        const int t = threadIdx.x;
        if (t < ADV) {  // cells a0+2 .. a0+241: 15 whole lines per row
#pragma unroll
            for (int ac = 0; ac < AC; ac++) put(HALO + t, ac);
        } else if (t < ADV + 2 * HALO) {  // the tile's own halo cells, only where they are ghost cells
            const int k = t - ADV;
            const int al = k < HALO ? k : ALONG - 2 * HALO + k;
            const int ga = a0 + al;
            if (ga < a.mbc || ga >= a.mbc + a.m_al)
                for (int ac = 0; ac < AC; ac++) put(al, ac);
        }
    } else {
#pragma unroll
        for (int k = 0; k < ALONG; k += 256 / AC) {
            const int al = k + threadIdx.x / AC;
            const int ga = a0 + al;
            const bool inner = (ga >= a.mbc) && (ga < a.mbc + a.m_al);
            if (inner ? (al >= HALO && al < ALONG - HALO) : true) put(al, threadIdx.x % AC);
        }
    }
    cfl_publish(a.cfl, cfl_value<CAPA>(cflmax, a.dtd));
}

}  // namespace PCL_NS
}  // namespace pcl
