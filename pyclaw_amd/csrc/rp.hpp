// rp.hpp -- pointwise Riemann solvers as device functions (gfx950).
//
// The reference's "vectorised" solvers loop over a slice and write slice arrays
// (rpn2(ixy,maxm,meqn,mwaves,mbc,mx,ql,qr,auxl,auxr,wave,s,amdq,apdq)); here one
// lane owns one interface and everything stays in registers.  Per-CELL quantities
// that the Fortran recomputes for the interface on either side of a cell
// (pressure, sqrt(rho), sound speed ...) are computed once per lane in precell()
// and handed to the right-hand neighbour with a wavefront shift.  Each expression
// keeps the reference's operation order so PCL_MATH_EXACT is bit-identical.
//
// Solver concept:
//   MEQN, MWAVES, NCELL (doubles in Cell)
//   template<int IXY> nz(mw,m)  -- compile-time sparsity of wave(m,mw)
//   template<int IXY> precell(q, par) -> Cell
//   template<int IXY> solve(L, R, par, wave, s, amdq, apdq)
//   template<int IXY> transverse(L, R, par, asdq, bmasdq, bpasdq)   (rpt2)
#pragma once
#include <hip/hip_runtime.h>
#include "sweep_args.hpp"

#ifndef PCL_NS
#error "rp.hpp is compiled once per arithmetic mode: define PCL_NS (exact|fast) and PCL_FAST (0|1)"
#endif

namespace pcl {
namespace PCL_NS {

// dmax1/dmin1 of the Fortran.  v_max_f64/v_min_f64 are one instruction; the compare+select
// form costs a v_cmp, a 2-wait-state VCC hazard and two v_cndmask.  They differ from the
// Fortran intrinsics only for NaN operands and in the sign of a zero result.
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }

// ---- square root ---------------------------------------------------------------------------
// Same Goldschmidt/Newton sequence hipcc emits for an IEEE f64 sqrt (v_rsq_f64 seed, one
// coupled g/h step, two fma residual corrections => correctly rounded), minus the 2^256 input
// pre-scaling it adds for arguments below 2^-767, which no density/sound-speed ever is.
// 0, +inf (and NaN / negative => NaN) behave like sqrt().
// PCL_FAST (rtol 1e-12 mode): one residual correction instead of two (2^-52 instead of correctly rounded).  A negative
// argument must still give NaN -- the entropy fix relies on "sound speed of an unphysical intermediate state is NaN,
// every comparison with it is false" (rpn2:217-243) -- so the seed is taken of x itself; only the zero is patched.
#ifndef PCL_FAST_SQRT          /* (separately switchable for A/B builds) */
#define PCL_FAST_SQRT PCL_FAST
#define PCL_FAST_SQRTH PCL_FAST
#define PCL_FAST_RECIP PCL_FAST
#endif
__device__ __forceinline__ double dsqrt(double x) {
#if PCL_FAST_SQRT
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;
#else
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return __builtin_amdgcn_class(x, 0x260) ? x : g;  // +-0, +inf pass through
#endif
}
// The same sequence without the +-0 / +inf pass-through (a v_cmp_class and two v_cndmask): for arguments that are
// strictly positive and finite in every physical state -- a density, the squared Roe sound speed.  (For 0 or inf
// it returns NaN where sqrt() returns 0 / inf: such a state is already outside the solver's domain, the very next
// operations divide by the root.)
// It also hands out the iteration's by-product h ~ 1/(2 sqrt x) (relative error <= 2^-47.8 measured over 10^6
// arguments, tools/ubench/seed_accuracy.hip): a far better reciprocal seed than v_rcp_f64 (2^-24), see
// Recip::seeded below.
struct SqrtH { double g, h; };
__device__ __forceinline__ SqrtH dsqrt_pos_h(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y;
    double h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
#if PCL_FAST_SQRTH
    return SqrtH{g, h};
#else
    d = __builtin_fma(-g, g, x);
    return SqrtH{__builtin_fma(d, h, g), h};
#endif
}
__device__ __forceinline__ double dsqrt_pos(double x) { return dsqrt_pos_h(x).g; }

// ---- division ------------------------------------------------------------------------------
// hipcc expands an IEEE f64 division into 2 v_div_scale, v_rcp_f64, 4 fma (Newton on the
// reciprocal), mul, fma, v_div_fmas, v_div_fixup: 11 instructions, 5 of them quarter-rate.
// Two observations make the Riemann solver's divisions much cheaper with THE SAME BITS:
//  * the reciprocal refinement depends on the denominator only, and the reference divides
//    several numerators by the same value (q/rho, q/sqrt(rho), ./rhsq2): refine once;
//  * v_div_scale / v_div_fmas / v_div_fixup only matter when an operand or the quotient is
//    outside the normal range (zero, inf, NaN, denormal, |exponent| > ~700).  Densities,
//    sound speeds and their sums never are, and for normal operands the plain sequence
//    q0 = n*r, q = fma(fma(-d,q0,n), r, q0) is the same correctly rounded quotient
//    (Markstein's theorem; r is 1/d to within an ulp after two Newton steps).
// So: Recip(d) = v_rcp_f64 + 4 fma, each quotient = mul + 2 fma.  Outside the normal range
// the result differs from IEEE: a zero / inf / NaN denominator (n/0 gives NaN instead of inf) is only
// reached by states that are already unphysical; a NUMERATOR in the underflow range (a momentum of 1e-310
// ahead of a front) is physical, and there the quotient can be one step of the denormal grid off -- see
// PCL_DENORM_GUARD below for the measured size of that and the build that removes it.  The limiter ratio (philim), whose operands are sums of
// squares that can legitimately underflow, keeps the full IEEE division in exact mode.
// PCL_DENORM_GUARD=1 (the `strict` build of the kernels, PCL_MATH_STRICT): quotients whose numerator lies in the underflow range take the IEEE division too.
// Not in the `exact` build: measured on MI355X it costs the VALU-bound kernels 3-5 % (dense state x +3.1 % / y +5.0 %, unsplit
// x phase +6.7 %; the memory-bound shock-bubble headline nothing), and what it buys is at most one step of the denormal
// grid per quotient: without it 952 of 5.1e6 values of tools/probe_denormal.py differ from the oracle, all below 1e-290
// in magnitude, by at most 3.2e-322; with it none do (profiles/r02_denorm_guard_ab.txt).  tests/test_gpu_fuzz.py pins
// that bound.
#ifndef PCL_DENORM_GUARD
#define PCL_DENORM_GUARD 0
#endif
struct Recip {
    double d, r;
    // From a seed y0 that is already 1/den to a few 2^-48 (a by-product of a square root of the same quantity, or
    // the square of such a reciprocal): ONE Newton step lands within an ulp, like the two steps from v_rcp_f64's
    // 2^-24 seed below -- no quarter-rate instruction, two fma instead of five instructions.  The quotients stay
    // correctly rounded for the same reason (Markstein; what matters is |r*den - 1| <~ 2^-52).
    __device__ __forceinline__ static Recip seeded(double den, double y0) {
        Recip x;
        x.d = den;
        const double e = __builtin_fma(-den, y0, 1.0);
        x.r = __builtin_fma(y0, e, y0);
        return x;
    }
    __device__ __forceinline__ Recip() {}
    __device__ __forceinline__ explicit Recip(double den) : d(den) {
        double y = __builtin_amdgcn_rcp(den);
        double e = __builtin_fma(-den, y, 1.0);
        y = __builtin_fma(y, e, y);
#if !PCL_FAST_RECIP   /* fast mode: one Newton step from the 2^-24 seed = 2^-48, plenty for rtol 1e-12 */
        e = __builtin_fma(-den, y, 1.0);
        y = __builtin_fma(y, e, y);
#endif
        r = y;
    }
    __device__ __forceinline__ double div(double n) const {
#if PCL_FAST
        return n * r;  // within ~1 ulp, not correctly rounded
#else
#if PCL_DENORM_GUARD
        // a numerator deep in the underflow range (|n| < 2^-960, not zero: the tail of a momentum or tracer field
        // ahead of a front): the residual below would itself underflow and the quotient could be off by one step of
        // the denormal grid -- take the IEEE division.  frexp_exp is 0 for n == 0, so quiescent regions stay on the
        // fast path; the empty asm keeps this a (practically never taken) branch instead of a select.
        if (__builtin_expect(__builtin_amdgcn_frexp_exp(n) < -960, 0)) {
            asm volatile("");
            return n / d;
        }
#endif
        const double q = n * r;
        return __builtin_fma(__builtin_fma(-d, q, n), r, q);
#endif
    }
};

// one division of physical (normal-range) quantities
__device__ __forceinline__ double fdiv(double n, double d) { return Recip(d).div(n); }

// one division with full IEEE semantics in exact mode (any operands)
__device__ __forceinline__ double fdiv_ieee(double n, double d) {
#if PCL_FAST
    return n * Recip(d).r;      // (the limiter clamps a vanishing |wave|^2 first: philim in classic.hpp)
#else
    return n / d;
#endif
}

// ------------------------------------------------------------------------------------
// 1-D advection, q_t + u q_x = 0  (third-party rp1_advection.f, restated)
// ------------------------------------------------------------------------------------
struct Advection1D {
    static constexpr int MEQN = 1, MWAVES = 1, NCELL = 1, NAUX = 0;
    struct Cell { double q[1]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &p,
                                                 double (&wave)[1][1], double (&s)[1],
                                                 double (&amdq)[1], double (&apdq)[1]) {
        const double u = p.v[0];
        wave[0][0] = R.q[0] - L.q[0];
        s[0] = u;
        amdq[0] = dmin(u, 0.0) * wave[0][0];
        apdq[0] = dmax(u, 0.0) * wave[0][0];
    }
    // wave speeds alone (what solve() would return in s), for wavefronts without any jump (classic.hpp)
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &, const RpParams &p, double (&s)[1]) {
        s[0] = p.v[0];
    }
};

// ------------------------------------------------------------------------------------
// 1-D colour equation q_t + u(x) q_x = 0 (third-party rp1_advection_color.f, restated): aux(1) of a cell is the
// velocity at its LEFT edge, so interface i uses the right cell's value
// ------------------------------------------------------------------------------------
struct AdvectionColor1D {
    static constexpr int MEQN = 1, MWAVES = 1, NCELL = 2, NAUX = 1, NAUX_T = 1;
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) { return k; }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) { return k; }
    struct Cell { double q[1]; double u; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &, const double *auxv) {
        Cell c; c.q[0] = q[0]; c.u = auxv[0]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[1][1], double (&s)[1],
                                                 double (&amdq)[1], double (&apdq)[1]) {
        wave[0][0] = R.q[0] - L.q[0];
        s[0] = R.u;
        amdq[0] = dmin(R.u, 0.0) * wave[0][0];
        apdq[0] = dmax(R.u, 0.0) * wave[0][0];
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &R, const RpParams &, double (&s)[1]) {
        s[0] = R.u;
    }
};

// ------------------------------------------------------------------------------------
// 1-D Burgers' equation with the transonic entropy fix (third-party rp1_burgers.f90, restated)
// ------------------------------------------------------------------------------------
struct Burgers1D {
    static constexpr int MEQN = 1, MWAVES = 1, NCELL = 1, NAUX = 0;
    struct Cell { double q[1]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[1][1], double (&s)[1],
                                                 double (&amdq)[1], double (&apdq)[1]) {
        wave[0][0] = R.q[0] - L.q[0];
        s[0] = 0.5 * (L.q[0] + R.q[0]);
        const bool transonic = R.q[0] > 0.0 && L.q[0] < 0.0;
        amdq[0] = transonic ? -0.5 * (L.q[0] * L.q[0]) : dmin(s[0], 0.0) * wave[0][0];
        apdq[0] = transonic ? 0.5 * (R.q[0] * R.q[0]) : dmax(s[0], 0.0) * wave[0][0];
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &, double (&s)[1]) {
        s[0] = 0.5 * (L.q[0] + R.q[0]);
    }
};

// ------------------------------------------------------------------------------------
// 1-D Euler equations, Roe solver + Harten-Hyman entropy fix (third-party rp1_euler_with_efix.f, restated:
// the vendored 2-D solver below without its shear and tracer waves); par = gamma, gamma1.
// Plain IEEE operations in the order of the C restatement (oracle/classic_oracle.c: rp1_euler).
// ------------------------------------------------------------------------------------
struct Euler1D {
    static constexpr int MEQN = 3, MWAVES = 3, NCELL = 3, NAUX = 0;
    struct Cell { double q[3]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; c.q[1] = q[1]; c.q[2] = q[2]; return c;
    }
    struct Roe { double u, enth, a2r, a, pl, pr; };
    __device__ static __forceinline__ Roe roe(const Cell &L, const Cell &R, double gamma1) {
        Roe r;
        const double rsl = dsqrt(L.q[0]), rsr = dsqrt(R.q[0]);
        r.pl = gamma1 * (L.q[2] - 0.5 * (L.q[1] * L.q[1]) / L.q[0]);
        r.pr = gamma1 * (R.q[2] - 0.5 * (R.q[1] * R.q[1]) / R.q[0]);
        const double rhsq2 = rsl + rsr;
        r.u = (L.q[1] / rsl + R.q[1] / rsr) / rhsq2;
        r.enth = ((L.q[2] + r.pl) / rsl + (R.q[2] + r.pr) / rsr) / rhsq2;
        r.a2r = gamma1 * (r.enth - .5 * (r.u * r.u));
        r.a = dsqrt(r.a2r);
        return r;
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &par, double (&s)[3]) {
        const Roe r = roe(L, R, par.v[1]);
        s[0] = r.u - r.a; s[1] = r.u; s[2] = r.u + r.a;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &par,
                                                 double (&wave)[3][3], double (&s)[3], double (&amdq)[3],
                                                 double (&apdq)[3]) {
        const double gamma = par.v[0], gamma1 = par.v[1];
        const Roe r = roe(L, R, gamma1);
        const double u = r.u, enth = r.enth, a = r.a;
        const double rl = L.q[0], ml = L.q[1], el = L.q[2], rr = R.q[0], mr = R.q[1], er = R.q[2];
        const double d1 = rr - rl, d2 = mr - ml, d3 = er - el;
        const double a2 = gamma1 / r.a2r * ((enth - u * u) * d1 + u * d2 - d3);
        const double a3 = (d2 + (a - u) * d1 - a * a2) / (2.0 * a);
        const double a1 = d1 - a2 - a3;
        wave[0][0] = a1; wave[0][1] = a1 * (u - a); wave[0][2] = a1 * (enth - u * a); s[0] = u - a;
        wave[1][0] = a2; wave[1][1] = a2 * u;       wave[1][2] = a2 * 0.5 * (u * u);  s[1] = u;
        wave[2][0] = a3; wave[2][1] = a3 * (u + a); wave[2][2] = a3 * (enth + u * a); s[2] = u + a;
        const double cl = dsqrt(gamma * r.pl / rl);
        const double s0 = ml / rl - cl;
        const bool all_right = (s0 >= 0.0) && (s[0] > 0.0);
        {
            const double rho1 = rl + wave[0][0], rhou1 = ml + wave[0][1], en1 = el + wave[0][2];
            const double p1 = gamma1 * (en1 - 0.5 * (rhou1 * rhou1) / rho1);
            const double c1 = dsqrt(gamma * p1 / rho1);
            const double s1 = rhou1 / rho1 - c1;
            double sfract;
            if (s0 < 0.0 && s1 > 0.0) sfract = s0 * (s1 - s[0]) / (s1 - s0);
            else if (s[0] < 0.0) sfract = s[0];
            else sfract = 0.0;
            for (int m = 0; m < 3; m++) amdq[m] = sfract * wave[0][m];
        }
        if (!(s[1] >= 0.0)) {
            for (int m = 0; m < 3; m++) amdq[m] = amdq[m] + s[1] * wave[1][m];
            const double cr = dsqrt(gamma * r.pr / rr);
            const double s3 = mr / rr + cr;
            const double rho2 = rr - wave[2][0], rhou2 = mr - wave[2][1], en2 = er - wave[2][2];
            const double p2 = gamma1 * (en2 - 0.5 * (rhou2 * rhou2) / rho2);
            const double c2 = dsqrt(gamma * p2 / rho2);
            const double s2 = rhou2 / rho2 + c2;
            double sfract = 0.0;
            bool add = true;
            if (s2 < 0.0 && s3 > 0.0) sfract = s2 * (s3 - s[2]) / (s3 - s2);
            else if (s[2] < 0.0) sfract = s[2];
            else add = false;
            if (add) for (int m = 0; m < 3; m++) amdq[m] = amdq[m] + sfract * wave[2][m];
        }
        if (all_right) for (int m = 0; m < 3; m++) amdq[m] = 0.0;
        for (int m = 0; m < 3; m++) {
            double df = s[0] * wave[0][m];
            df = df + s[1] * wave[1][m];
            df = df + s[2] * wave[2][m];
            apdq[m] = df - amdq[m];
        }
    }
};

// ------------------------------------------------------------------------------------
// 1-D shallow water, Roe solver + Harten-Hyman entropy fix (third-party rp1_shallow_roe_with_efix.f, restated);
// q = (h, hu); par = g.  Same operation order as oracle/classic_oracle.c: rp1_shallow.
// ------------------------------------------------------------------------------------
struct Shallow1D {
    static constexpr int MEQN = 2, MWAVES = 2, NCELL = 2, NAUX = 0;
    struct Cell { double q[2]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; c.q[1] = q[1]; return c;
    }
    __device__ static __forceinline__ void roe(const Cell &L, const Cell &R, double g, double &ubar, double &cbar) {
        const double hsl = dsqrt(L.q[0]), hsr = dsqrt(R.q[0]);
        ubar = (L.q[1] / hsl + R.q[1] / hsr) / (hsl + hsr);
        cbar = dsqrt(0.5 * g * (L.q[0] + R.q[0]));
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &par, double (&s)[2]) {
        double ubar, cbar;
        roe(L, R, par.v[0], ubar, cbar);
        s[0] = ubar - cbar; s[1] = ubar + cbar;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &par,
                                                 double (&wave)[2][2], double (&s)[2], double (&amdq)[2],
                                                 double (&apdq)[2]) {
        const double g = par.v[0];
        const double hl = L.q[0], ml = L.q[1], hr = R.q[0], mr = R.q[1];
        double ubar, cbar;
        roe(L, R, g, ubar, cbar);
        const double d1 = hr - hl, d2 = mr - ml;
        const double a1 = 0.5 * (-d2 + (ubar + cbar) * d1) / cbar;
        const double a2 = 0.5 * (d2 - (ubar - cbar) * d1) / cbar;
        wave[0][0] = a1; wave[0][1] = a1 * (ubar - cbar); s[0] = ubar - cbar;
        wave[1][0] = a2; wave[1][1] = a2 * (ubar + cbar); s[1] = ubar + cbar;
        const double s0 = ml / hl - dsqrt(g * hl);
        const bool all_right = (s0 >= 0.0) && (s[0] > 0.0);
        {
            const double h1 = hl + wave[0][0], hu1 = ml + wave[0][1];
            const double s1 = hu1 / h1 - dsqrt(g * h1);
            double sfract;
            if (s0 < 0.0 && s1 > 0.0) sfract = s0 * (s1 - s[0]) / (s1 - s0);
            else if (s[0] < 0.0) sfract = s[0];
            else sfract = 0.0;
            for (int m = 0; m < 2; m++) amdq[m] = sfract * wave[0][m];
            const double s3 = mr / hr + dsqrt(g * hr);
            const double h2 = hr - wave[1][0], hu2 = mr - wave[1][1];
            const double s2 = hu2 / h2 + dsqrt(g * h2);
            bool add = true;
            if (s2 < 0.0 && s3 > 0.0) sfract = s2 * (s3 - s[1]) / (s3 - s2);
            else if (s[1] < 0.0) sfract = s[1];
            else add = false;
            if (add) for (int m = 0; m < 2; m++) amdq[m] = amdq[m] + sfract * wave[1][m];
        }
        if (all_right) for (int m = 0; m < 2; m++) amdq[m] = 0.0;
        for (int m = 0; m < 2; m++) {
            double df = s[0] * wave[0][m];
            df = df + s[1] * wave[1][m];
            apdq[m] = df - amdq[m];
        }
    }
};

// ------------------------------------------------------------------------------------
// 2-D constant-coefficient advection (third-party rpn2_advection.f / rpt2_advection.f, restated); par = u, v
// ------------------------------------------------------------------------------------
struct Advection2D {
    static constexpr int MEQN = 1, MWAVES = 1, NCELL = 1, NAUX = 0;
    struct Cell { double q[1]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &p,
                                                 double (&wave)[1][1], double (&s)[1],
                                                 double (&amdq)[1], double (&apdq)[1]) {
        const double vel = p.v[IXY - 1];
        wave[0][0] = R.q[0] - L.q[0];
        s[0] = vel;
        amdq[0] = dmin(vel, 0.0) * wave[0][0];
        apdq[0] = dmax(vel, 0.0) * wave[0][0];
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &, const RpParams &p, double (&s)[1]) {
        s[0] = p.v[IXY - 1];
    }
    template <int IXY>
    __device__ static __forceinline__ void transverse(const Cell &, const Cell &, const RpParams &p,
                                                      const double (&asdq)[1], double (&bm)[1], double (&bp)[1]) {
        const double stran = p.v[2 - IXY];
        bm[0] = dmin(stran, 0.0) * asdq[0];
        bp[0] = dmax(stran, 0.0) * asdq[0];
    }
};

// ------------------------------------------------------------------------------------
// 1-D acoustics (third-party rp1_acoustics.f, restated); par = rho,bulk,cc,zz
// ------------------------------------------------------------------------------------
struct Acoustics1D {
    static constexpr int MEQN = 2, MWAVES = 2, NCELL = 2, NAUX = 0;
    struct Cell { double q[2]; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; c.q[1] = q[1]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &p,
                                                 double (&wave)[2][2], double (&s)[2],
                                                 double (&amdq)[2], double (&apdq)[2]) {
        const double cc = p.v[2], zz = p.v[3];
        const double d1 = R.q[0] - L.q[0];
        const double d2 = R.q[1] - L.q[1];
        const double a1 = (-d1 + zz * d2) / (2.0 * zz);
        const double a2 = (d1 + zz * d2) / (2.0 * zz);
        wave[0][0] = -a1 * zz; wave[0][1] = a1; s[0] = -cc;
        wave[1][0] = a2 * zz;  wave[1][1] = a2; s[1] = cc;
        for (int m = 0; m < 2; m++) { amdq[m] = s[0] * wave[0][m]; apdq[m] = s[1] * wave[1][m]; }
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &, const RpParams &p, double (&s)[2]) {
        s[0] = -p.v[2]; s[1] = p.v[2];
    }
};

// ------------------------------------------------------------------------------------
// 2-D acoustics (third-party rpn2_acoustics.f / rpt2_acoustics.f, restated)
// q = (p, u, v); par = rho,bulk,cc,zz
// ------------------------------------------------------------------------------------
struct Acoustics2D {
    static constexpr int MEQN = 3, MWAVES = 2, NCELL = 3, NAUX = 0;
    struct Cell { double q[3]; };
    template <int IXY> __device__ static constexpr bool nz(int /*mw*/, int m) {
        return m == 0 || m == (IXY == 1 ? 1 : 2);
    }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; c.q[1] = q[1]; c.q[2] = q[2]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &p,
                                                 double (&wave)[2][3], double (&s)[2],
                                                 double (&amdq)[3], double (&apdq)[3]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double cc = p.v[2], zz = p.v[3];
        const double d1 = R.q[0] - L.q[0];
        const double d2 = R.q[mu] - L.q[mu];
        const double a1 = (-d1 + zz * d2) / (2.0 * zz);
        const double a2 = (d1 + zz * d2) / (2.0 * zz);
        wave[0][0] = -a1 * zz; wave[0][mu] = a1; wave[0][mv] = 0.0; s[0] = -cc;
        wave[1][0] = a2 * zz;  wave[1][mu] = a2; wave[1][mv] = 0.0; s[1] = cc;
        for (int m = 0; m < 3; m++) { amdq[m] = s[0] * wave[0][m]; apdq[m] = s[1] * wave[1][m]; }
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &, const RpParams &p, double (&s)[2]) {
        s[0] = -p.v[2]; s[1] = p.v[2];
    }
    template <int IXY>
    __device__ static __forceinline__ void transverse(const Cell &, const Cell &, const RpParams &p,
                                                      const double (&asdq)[3], double (&bm)[3],
                                                      double (&bp)[3]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double cc = p.v[2], zz = p.v[3];
        const double a1 = (-asdq[0] + zz * asdq[mv]) / (2.0 * zz);
        const double a2 = (asdq[0] + zz * asdq[mv]) / (2.0 * zz);
        bm[0] = cc * a1 * zz; bm[mu] = 0.0; bm[mv] = -cc * a1;
        bp[0] = cc * a2 * zz; bp[mu] = 0.0; bp[mv] = cc * a2;
    }
};

// ---- 2-D colour equation with edge velocities (third-party rpn2_vc_advection.f / rpt2_vc_advection.f, restated):
// aux(1) = u at the cell's left edge, aux(2) = v at its bottom edge
struct VcAdvection2D {
    static constexpr int MEQN = 1, MWAVES = 1, NAUX = 2, NAUX_T = 2;
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) { return k; }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) { return k; }
    struct Cell { double q[1]; double u, v; };
    template <int IXY> __device__ static constexpr bool nz(int, int) { return true; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &, const double *auxv) {
        Cell c; c.q[0] = q[0]; c.u = auxv[0]; c.v = auxv[1]; return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[1][1], double (&s)[1], double (&amdq)[1],
                                                 double (&apdq)[1]) {
        const double vel = IXY == 1 ? R.u : R.v;
        wave[0][0] = R.q[0] - L.q[0];
        s[0] = vel;
        amdq[0] = dmin(vel, 0.0) * wave[0][0];
        apdq[0] = dmax(vel, 0.0) * wave[0][0];
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &, const Cell &R, const RpParams &, double (&s)[1]) {
        s[0] = IXY == 1 ? R.u : R.v;
    }
    // down-going part: the transverse velocity at this cell's own lower edge; up-going: at the lower edge of the
    // cell above (auxa)
    template <int IXY>
    __device__ static __forceinline__ void transverse_vc(const Cell &c1, const double * /*auxo*/, const double * /*auxb*/,
                                                         const double *auxa, const RpParams &,
                                                         const double (&asdq)[1], double (&bm)[1], double (&bp)[1]) {
        const double own = IXY == 1 ? c1.v : c1.u, above = IXY == 1 ? auxa[1] : auxa[0];
        bm[0] = dmin(own, 0.0) * asdq[0];
        bp[0] = dmax(above, 0.0) * asdq[0];
    }
};

// ---- 2-D acoustics with cell-wise impedance and sound speed (third-party rpn2_vc_acoustics.f, restated: the
// formulas of VcAcoustics3D below); q = (p, u, v); aux(1) = Z, aux(2) = c.  The transverse solver
// (rpt2_vc_acoustics.f, restated) needs the aux values of the two neighbouring slices: transverse_vc.
struct VcAcoustics2D {
    static constexpr int MEQN = 3, MWAVES = 2, NAUX = 2, NAUX_T = 2;
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) { return k; }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) { return k; }
    struct Cell { double q[3]; double z, c; };
    template <int IXY> __device__ static constexpr bool nz(int /*mw*/, int m) { return m == 0 || m == IXY; }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &, const double *auxv) {
        Cell c;
        for (int m = 0; m < 3; m++) c.q[m] = q[m];
        c.z = auxv[0];
        c.c = auxv[1];
        return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[2][3], double (&s)[2], double (&amdq)[3],
                                                 double (&apdq)[3]) {
        constexpr int mu = IXY;
        const double d1 = R.q[0] - L.q[0];
        const double d2 = R.q[mu] - L.q[mu];
        const double zi = R.z, zim = L.z;
        const Recip by_zz(zim + zi);
        const double a1 = by_zz.div(-d1 + zi * d2);
        const double a2 = by_zz.div(d1 + zim * d2);
        for (int m = 0; m < 3; m++) { wave[0][m] = 0.0; wave[1][m] = 0.0; }
        wave[0][0] = -a1 * zim; wave[0][mu] = a1; s[0] = -L.c;
        wave[1][0] = a2 * zi;   wave[1][mu] = a2; s[1] = R.c;
        for (int m = 0; m < 3; m++) {
            amdq[m] = nz<IXY>(0, m) ? s[0] * wave[0][m] : 0.0;
            apdq[m] = nz<IXY>(1, m) ? s[1] * wave[1][m] : 0.0;
        }
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &, double (&s)[2]) {
        s[0] = -L.c; s[1] = R.c;
    }
    // c1 = the cell asdq belongs to (left of the interface for amdq, right for apdq); auxb / auxa = aux of the
    // cells below / above it in the transverse direction.  The down-going part enters the slice below with that
    // slice's impedance and sound speed, the up-going part the slice above.
    template <int IXY>
    __device__ static __forceinline__ void transverse_vc(const Cell &c1, const double * /*auxo*/, const double *auxb,
                                                         const double *auxa, const RpParams &,
                                                         const double (&asdq)[3], double (&bm)[3], double (&bp)[3]) {
        constexpr int mu = IXY, mv = (IXY == 1) ? 2 : 1;
        const double zm = auxb[0], zz = c1.z, zp = auxa[0], cm = auxb[1], cp = auxa[1];
        const double a1 = fdiv_ieee(-asdq[0] + asdq[mv] * zz, zm + zz);
        const double a2 = fdiv_ieee(asdq[0] + asdq[mv] * zz, zz + zp);
        bm[0] = cm * a1 * zm; bm[mu] = 0.0; bm[mv] = -cm * a1;
        bp[0] = cp * a2 * zp; bp[mu] = 0.0; bp[mv] = cp * a2;
    }
};

// ---- 3-D acoustics with cell-wise impedance and sound speed (third-party rpn3_vc_acoustics.f, restated;
// named by the reference's test/acoustics/3d/Makefile) --------------------------------------------------
// q = (p, u, v, w); aux(1) = Z, aux(2) = c.  DIR = 1,2,3 selects the normal velocity q(DIR).
struct VcAcoustics3D {
    static constexpr int MEQN = 4, MWAVES = 2, NAUX = 2, NAUX_T = 2;
    // the transverse solvers are driven by the pressure component alone (classic3.hpp: slice3_pieces_p)
    static constexpr bool T3_PRESSURE = true;
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) { return k; }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) { return k; }
    struct Cell { double q[4]; double z, c; };
    template <int DIR> __device__ static constexpr bool nz(int /*mw*/, int m) { return m == 0 || m == DIR; }
    template <int DIR>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &, const double *auxv) {
        Cell c;
        for (int m = 0; m < 4; m++) c.q[m] = q[m];
        c.z = auxv[0];
        c.c = auxv[1];
        return c;
    }
    template <int DIR>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[2][4], double (&s)[2], double (&amdq)[4],
                                                 double (&apdq)[4]) {
        constexpr int mu = DIR;
        const double d1 = R.q[0] - L.q[0];
        const double d2 = R.q[mu] - L.q[mu];
        const double zi = R.z, zim = L.z;
        const Recip by_zz(zim + zi);
        const double a1 = by_zz.div(-d1 + zi * d2);
        const double a2 = by_zz.div(d1 + zim * d2);
        for (int m = 0; m < 4; m++) { wave[0][m] = 0.0; wave[1][m] = 0.0; }
        wave[0][0] = -a1 * zim; wave[0][mu] = a1; s[0] = -L.c;
        wave[1][0] = a2 * zi;   wave[1][mu] = a2; s[1] = R.c;
        for (int m = 0; m < 4; m++) {
            amdq[m] = nz<DIR>(0, m) ? s[0] * wave[0][m] : 0.0;
            apdq[m] = nz<DIR>(1, m) ? s[1] * wave[1][m] : 0.0;
        }
    }
    template <int DIR>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &, double (&s)[2]) {
        s[0] = -L.c; s[1] = R.c;
    }
    // Transverse solvers of the unsplit algorithm (third-party rpt3_vc_acoustics.f / rptt3_vc_acoustics.f, restated:
    // oracle/classic_oracle.c).  blk[oe+1][of+1][k]: aux component k (0 = Z, 1 = c) of the cell asdq sits in and of its
    // neighbours at y-like offset oe / z-like offset of.  icoor = 2 splits in the y-like, 3 in the z-like direction.
    __device__ static __forceinline__ double pick(const double (&a)[4], int iuvw) {
        return iuvw == 1 ? a[1] : (iuvw == 2 ? a[2] : a[3]);
    }
    // The 16 transverse solves of one interface divide by sums of neighbouring impedances, and only twelve different
    // sums occur in a cell's 3 x 3 block: the pairs (lower, centre) and (centre, upper) of its three lines along the
    // y-like and its three lines along the z-like direction.  BlkRcp holds their reciprocals (refined once; every
    // quotient is then the correctly rounded Markstein form of Recip::div, like the normal solve's) -- 12 reciprocals
    // per lane instead of 32 IEEE divisions, and the left neighbour's set arrives by DPP shift.
    struct BlkRcp {
        Recip y[3][2];     // y[r][0|1]: line along the y-like direction at z-like offset r-1: Z(-1)+Z(0), Z(0)+Z(+1)
        Recip z[3][2];     // z[r][0|1]: line along the z-like direction at y-like offset r-1
    };
    __device__ static __forceinline__ BlkRcp blk_rcp(const double (&blk)[3][3][2]) {
        BlkRcp b;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            b.y[r][0] = Recip(blk[0][r][0] + blk[1][r][0]); b.y[r][1] = Recip(blk[1][r][0] + blk[2][r][0]);
            b.z[r][0] = Recip(blk[r][0][0] + blk[r][1][0]); b.z[r][1] = Recip(blk[r][1][0] + blk[r][2][0]);
        }
        return b;
    }
    template <int DIR>
    __device__ static __forceinline__ void transverse3(int icoor, const double (&blk)[3][3][2], const BlkRcp &rc,
                                                       const double (&asdq)[4], double (&bm)[4], double (&bp)[4]) {
        int iuvw = DIR + icoor - 1;
        if (iuvw > 3) iuvw -= 3;
        double t[4] = {asdq[0], pick(asdq, iuvw), 0.0, 0.0};
        double om[4], op[4];
        if (icoor == 2)
            split3x(blk[0][1][0], blk[1][1][0], blk[2][1][0], blk[0][1][1], blk[2][1][1], rc.y[1][0], rc.y[1][1], t, om, op);
        else
            split3x(blk[1][0][0], blk[1][1][0], blk[1][2][0], blk[1][0][1], blk[1][2][1], rc.z[1][0], rc.z[1][1], t, om, op);
        place(iuvw, om, op, bm, bp);
    }
    template <int DIR>
    __device__ static __forceinline__ void transverse3t(int icoor, int impt, const double (&blk)[3][3][2], const BlkRcp &rc,
                                                        const double (&bsasdq)[4], double (&cmo)[4], double (&cpo)[4]) {
        int iuvw = DIR + icoor - 1;
        if (iuvw > 3) iuvw -= 3;
        double t[4] = {bsasdq[0], pick(bsasdq, iuvw), 0.0, 0.0};
        double om[4], op[4];
        const int r = impt == 1 ? 0 : 2;
        if (icoor == 2)       // new split in the y-like direction, inside the z-like row the first split went to
            split3x(sel(blk, 0, r, 0), sel(blk, 1, r, 0), sel(blk, 2, r, 0), sel(blk, 0, r, 1), sel(blk, 2, r, 1),
                    r == 0 ? rc.y[0][0] : rc.y[2][0], r == 0 ? rc.y[0][1] : rc.y[2][1], t, om, op);
        else                  // new split in the z-like direction, inside the y-like row the first split went to
            split3x(sel(blk, r, 0, 0), sel(blk, r, 1, 0), sel(blk, r, 2, 0), sel(blk, r, 0, 1), sel(blk, r, 2, 1),
                    r == 0 ? rc.z[0][0] : rc.z[2][0], r == 0 ? rc.z[0][1] : rc.z[2][1], t, om, op);
        place(iuvw, om, op, cmo, cpo);
    }
    // blk[a][b][k] with a or b in {0, 2} chosen at run time, written with static indices
    __device__ static __forceinline__ double sel(const double (&blk)[3][3][2], int a, int b, int k) {
        double v = 0.0;
#pragma unroll
        for (int x = 0; x < 3; x++)
#pragma unroll
            for (int y = 0; y < 3; y++)
#pragma unroll
                for (int z = 0; z < 2; z++)
                    v = (x == a && y == b && z == k) ? blk[x][y][z] : v;
        return v;
    }
    // t = (asdq(1), asdq(iuvw+1)); om/op = (pressure part, velocity part)
    __device__ static __forceinline__ void split3x(double zm, double zz, double zp, double cm, double cp,
                                                   const Recip &by_m, const Recip &by_p,        // 1/(zm+zz), 1/(zz+zp)
                                                   const double (&t)[4], double (&om)[4], double (&op)[4]) {
        const double a1 = by_m.div(-t[0] + t[1] * zz);
        const double a2 = by_p.div(t[0] + t[1] * zz);
        om[0] = cm * a1 * zm; om[1] = -cm * a1; om[2] = om[3] = 0.0;
        op[0] = cp * a2 * zp; op[1] = cp * a2; op[2] = op[3] = 0.0;
    }
    __device__ static __forceinline__ void place(int iuvw, const double (&om)[4], const double (&op)[4],
                                                 double (&bm)[4], double (&bp)[4]) {
        bm[0] = om[0]; bp[0] = op[0];
        bm[1] = iuvw == 1 ? om[1] : 0.0; bp[1] = iuvw == 1 ? op[1] : 0.0;
        bm[2] = iuvw == 2 ? om[1] : 0.0; bp[2] = iuvw == 2 ? op[1] : 0.0;
        bm[3] = iuvw == 3 ? om[1] : 0.0; bp[3] = iuvw == 3 ? op[1] : 0.0;
    }
};

// ------------------------------------------------------------------------------------
// 2-D shallow water, Roe solver + Harten-Hyman entropy fix and its transverse solver (third-party
// rpn2_shallow_roe_with_efix.f / rpt2_shallow_roe_with_efix.f, restated); q = (h, hu, hv); par = g.
// Plain IEEE operations in the order of oracle/classic_oracle.c: rpn2_shallow / rpt2_shallow.
// ------------------------------------------------------------------------------------
struct Shallow2D {
    static constexpr int MEQN = 3, MWAVES = 3, NCELL = 3, NAUX = 0;
    struct Cell { double q[3]; };
    // wave(m,mw) sparsity: wave 2 carries only the transverse momentum
    template <int IXY> __device__ static constexpr bool nz(int mw, int m) {
        constexpr int mv = (IXY == 1) ? 2 : 1;
        return mw == 1 ? (m == mv) : true;
    }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &) {
        Cell c; c.q[0] = q[0]; c.q[1] = q[1]; c.q[2] = q[2]; return c;
    }
    struct Roe { double h, u, v, a; };
    template <int IXY> __device__ static __forceinline__ Roe roe(const Cell &L, const Cell &R, double g) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        Roe r;
        r.h = (L.q[0] + R.q[0]) * 0.5;
        const double hsl = dsqrt(L.q[0]), hsr = dsqrt(R.q[0]), hsq2 = hsl + hsr;
        r.u = (L.q[mu] / hsl + R.q[mu] / hsr) / hsq2;
        r.v = (L.q[mv] / hsl + R.q[mv] / hsr) / hsq2;
        r.a = dsqrt(g * r.h);
        return r;
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &par, double (&s)[3]) {
        const Roe r = roe<IXY>(L, R, par.v[0]);
        s[0] = r.u - r.a; s[1] = r.u; s[2] = r.u + r.a;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &par,
                                                 double (&wave)[3][3], double (&s)[3], double (&amdq)[3],
                                                 double (&apdq)[3]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double g = par.v[0];
        const Roe r = roe<IXY>(L, R, g);
        const double d1 = R.q[0] - L.q[0], d2 = R.q[mu] - L.q[mu], d3 = R.q[mv] - L.q[mv];
        const double a1 = ((r.u + r.a) * d1 - d2) * (0.5 / r.a);
        const double a2 = -r.v * d1 + d3;
        const double a3 = (-(r.u - r.a) * d1 + d2) * (0.5 / r.a);
        wave[0][0] = a1;  wave[0][mu] = a1 * (r.u - r.a); wave[0][mv] = a1 * r.v; s[0] = r.u - r.a;
        wave[1][0] = 0.0; wave[1][mu] = 0.0;              wave[1][mv] = a2;       s[1] = r.u;
        wave[2][0] = a3;  wave[2][mu] = a3 * (r.u + r.a); wave[2][mv] = a3 * r.v; s[2] = r.u + r.a;
        const double hl = L.q[0], hr = R.q[0];
        const double s0 = L.q[mu] / hl - dsqrt(g * hl);
        const bool all_right = (s0 >= 0.0) && (s[0] > 0.0);
        {
            const double h1 = hl + wave[0][0], hu1 = L.q[mu] + wave[0][mu];
            const double s1 = hu1 / h1 - dsqrt(g * h1);
            double sfract;
            if (s0 < 0.0 && s1 > 0.0) sfract = s0 * (s1 - s[0]) / (s1 - s0);
            else if (s[0] < 0.0) sfract = s[0];
            else sfract = 0.0;
            for (int m = 0; m < 3; m++) amdq[m] = sfract * wave[0][m];
        }
        if (!(s[1] >= 0.0)) {
            for (int m = 0; m < 3; m++) amdq[m] = amdq[m] + s[1] * wave[1][m];
            const double s03 = R.q[mu] / hr + dsqrt(g * hr);
            const double h3 = hr - wave[2][0], hu3 = R.q[mu] - wave[2][mu];
            const double s3 = hu3 / h3 + dsqrt(g * h3);
            double sfract = 0.0;
            bool add = true;
            if (s3 < 0.0 && s03 > 0.0) sfract = s3 * (s03 - s[2]) / (s03 - s3);
            else if (s[2] < 0.0) sfract = s[2];
            else add = false;
            if (add) for (int m = 0; m < 3; m++) amdq[m] = amdq[m] + sfract * wave[2][m];
        }
        if (all_right) for (int m = 0; m < 3; m++) amdq[m] = 0.0;
        for (int m = 0; m < 3; m++) {
            double df = s[0] * wave[0][m];
            df = df + s[1] * wave[1][m];
            df = df + s[2] * wave[2][m];
            apdq[m] = df - amdq[m];
        }
    }
    template <int IXY>
    __device__ static __forceinline__ void transverse(const Cell &L, const Cell &R, const RpParams &par,
                                                      const double (&asdq)[3], double (&bm)[3], double (&bp)[3]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const Roe r = roe<IXY>(L, R, par.v[0]);
        const double a1 = (0.5 / r.a) * ((r.v + r.a) * asdq[0] - asdq[mv]);
        const double a2 = asdq[mu] - r.u * asdq[0];
        const double a3 = (0.5 / r.a) * (-(r.v - r.a) * asdq[0] + asdq[mv]);
        double wb[3][3], sb[3];
        wb[0][0] = a1;  wb[0][mu] = a1 * r.u; wb[0][mv] = a1 * (r.v - r.a); sb[0] = r.v - r.a;
        wb[1][0] = 0.0; wb[1][mu] = a2;       wb[1][mv] = 0.0;              sb[1] = r.v;
        wb[2][0] = a3;  wb[2][mu] = a3 * r.u; wb[2][mv] = a3 * (r.v + r.a); sb[2] = r.v + r.a;
        for (int m = 0; m < 3; m++) {
            double m_ = 0.0, p_ = 0.0;
            for (int mw = 0; mw < 3; mw++) {
                m_ = m_ + dmin(sb[mw], 0.0) * wb[mw][m];
                p_ = p_ + dmax(sb[mw], 0.0) * wb[mw][m];
            }
            bm[m] = m_; bp[m] = p_;
        }
    }
};



// ------------------------------------------------------------------------------------
// f-wave solvers (FWAVE kernels: flux2fw.f / step1fw.f): nonlinear elasticity in 1-D (third-party
// rp1_nonlinear_elasticity_fwave.f, apps/elasticity/1d/stegoton) and the p-system in 2-D (third-party rpn2_psystem.f /
// rpt2_psystem.f, test/psystem), restated; same operation order as oracle/classic_oracle.c: rp_fwave_normal /
// rpt2_psystem.  aux(1) = rho, aux(2) = K, aux(3) = 1: sigma = K eps, else sigma = exp(K eps) - 1; the p-system's
// transverse solver also reads aux(4) = eps of the neighbouring rows.  wave[][] holds F-WAVES.
// (exp() is the device library's: the exponential law agrees with the CPU oracle to an ulp, not bit for bit.)
// ------------------------------------------------------------------------------------
__device__ __forceinline__ double ps_sigma(double eps, double K, double lin) { return lin == 1.0 ? K * eps : exp(K * eps) - 1.0; }
__device__ __forceinline__ double ps_sigmap(double eps, double K, double lin) { return lin == 1.0 ? K : K * exp(K * eps); }

template <int NEQ> struct FwaveElastic {
    static constexpr int MEQN = NEQ, MWAVES = 2, NAUX = 3, NAUX_T = 4;
    static constexpr bool IS_FWAVE = true;
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) { return k; }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) { return k; }
    struct Cell { double q[NEQ]; double rho, un, sig, c, z; };
    template <int IXY> __device__ static constexpr int mu_of() { return NEQ == 2 ? 1 : IXY; }
    template <int IXY> __device__ static constexpr bool nz(int /*mw*/, int m) { return m == 0 || m == mu_of<IXY>(); }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &, const double *auxv) {
        Cell c;
        for (int m = 0; m < NEQ; m++) c.q[m] = q[m];
        c.rho = auxv[0];
        c.un = fdiv_ieee(q[mu_of<IXY>()], c.rho);
        c.sig = ps_sigma(q[0], auxv[1], auxv[2]);
        c.c = dsqrt(fdiv_ieee(ps_sigmap(q[0], auxv[1], auxv[2]), c.rho));
        c.z = c.c * c.rho;
        return c;
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &, double (&s)[2]) {
        s[0] = -L.c; s[1] = R.c;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &,
                                                 double (&wave)[2][NEQ], double (&s)[2], double (&amdq)[NEQ],
                                                 double (&apdq)[NEQ]) {
        constexpr int mu = mu_of<IXY>();
        const double du = R.un - L.un;
        const double dsig = R.sig - L.sig;
        const double zs = L.z + R.z;
        const double b1 = fdiv_ieee(-(R.z * du + dsig), zs);
        const double b2 = fdiv_ieee(-(L.z * du - dsig), zs);
        for (int m = 0; m < NEQ; m++) { wave[0][m] = 0.0; wave[1][m] = 0.0; }
        wave[0][0] = b1; wave[0][mu] = b1 * L.z; s[0] = -L.c;
        wave[1][0] = b2; wave[1][mu] = b2 * (-R.z); s[1] = R.c;
        for (int m = 0; m < NEQ; m++) { amdq[m] = wave[0][m]; apdq[m] = wave[1][m]; }
    }
    // p-system only (NEQ == 3): auxb / auxa = (rho, K, flag, eps) of the rows below / above the cell asdq sits in
    template <int IXY>
    __device__ static __forceinline__ void transverse_vc(const Cell &, const double * /*auxo*/, const double *auxb,
                                                         const double *auxa, const RpParams &,
                                                         const double (&asdq)[NEQ], double (&bm)[NEQ], double (&bp)[NEQ]) {
        constexpr int mu = mu_of<IXY>(), mv = NEQ == 3 ? 3 - mu : 0;
        const double cm = dsqrt(fdiv_ieee(ps_sigmap(auxb[3], auxb[1], auxb[2]), auxb[0]));
        const double cp = dsqrt(fdiv_ieee(ps_sigmap(auxa[3], auxa[1], auxa[2]), auxa[0]));
        const double zm = cm * auxb[0], zp = cp * auxa[0];
        const double b1 = fdiv_ieee(asdq[mv] + zp * asdq[0], zm + zp);
        const double b3 = fdiv_ieee(zm * asdq[0] - asdq[mv], zm + zp);
        for (int m = 0; m < NEQ; m++) { bm[m] = 0.0; bp[m] = 0.0; }
        bm[0] = -cm * b1; bm[mv] = -cm * b1 * zm;
        bp[0] = cp * b3;  bp[mv] = cp * b3 * (-zp);
    }
};
using Elasticity1D = FwaveElastic<2>;
using PSystem2D = FwaveElastic<3>;

// ------------------------------------------------------------------------------------
// Shallow water on the sphere (Calhoun, Helzel & LeVeque 2008), q = (h, hu, hv, hw) with Cartesian momentum;
// third-party rpn2_shallow_sphere.f / rpt2_shallow_sphere.f (test/shallow_sphere/Makefile:7), restated; same
// operation order as oracle/classic_oracle.c: rpn2_sphere / rpt2_sphere / sphere_qcor.  par = g, dxcom, dycom.
// aux (setaux.f:10-25): 0 kappa, 1-3 / 4-6 normal and tangent of the LEFT edge, 7-9 / 10-12 of the BOTTOM edge,
// 13-15 radial unit vector at the cell centre.  A sweep along IXY needs the edge of that direction + the radial
// vector (9 components: aux_index); the transverse solves need the OTHER direction's edge + radial vector of the
// cell itself and of its two neighbours across (auxt_index).
// ------------------------------------------------------------------------------------
struct ShallowSphere {
    static constexpr int MEQN = 4, MWAVES = 3, NAUX = 9, NAUX_T = 9;
    static constexpr bool HAS_QCOR = true;   // the app replaces step2.f by step2qcor.f (Makefile:16)
    template <int IXY> __host__ __device__ static constexpr int aux_index(int k) {
        return k < 6 ? (IXY == 1 ? 1 + k : 7 + k) : 13 + (k - 6);
    }
    template <int IXY> __host__ __device__ static constexpr int auxt_index(int k) {
        return k < 6 ? (IXY == 1 ? 7 + k : 1 + k) : 13 + (k - 6);
    }
    struct Cell {
        double q[4];
        double er[3];          // radial unit vector of the cell
        double hs, sgh;        // sqrt(h), sqrt(g*h)
        double en[3], et[3];   // unit normal / unit tangent of the cell's own left (IXY 1) or bottom (IXY 2) edge
        double gam;            // length of that edge
    };
    template <int IXY> __device__ static constexpr bool nz(int mw, int m) { return !(mw == 1 && m == 0); }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &par, const double *auxv) {
        Cell c;
        for (int m = 0; m < 4; m++) c.q[m] = q[m];
        for (int k = 0; k < 3; k++) { c.en[k] = auxv[k]; c.er[k] = auxv[6 + k]; }
        const double etx = auxv[3], ety = auxv[4], etz = auxv[5];
        c.gam = dsqrt(etx * etx + ety * ety + etz * etz);
        const Recip by_gam(c.gam);
        c.et[0] = by_gam.div(etx); c.et[1] = by_gam.div(ety); c.et[2] = by_gam.div(etz);
        c.hs = dsqrt(q[0]);
        c.sgh = dsqrt(par.v[0] * q[0]);
        return c;
    }
    struct Roe { double u, v, a, hunl, hunr, hutl, hutr; };
    // L = qr(i-1) (the reference's suffix "r"), R = ql(i) (suffix "l"); the edge belongs to R
    __device__ static __forceinline__ Roe roe(const Cell &L, const Cell &R, double g) {
        Roe r;
        r.hunl = R.en[0] * R.q[1] + R.en[1] * R.q[2] + R.en[2] * R.q[3];
        r.hunr = R.en[0] * L.q[1] + R.en[1] * L.q[2] + R.en[2] * L.q[3];
        r.hutl = R.et[0] * R.q[1] + R.et[1] * R.q[2] + R.et[2] * R.q[3];
        r.hutr = R.et[0] * L.q[1] + R.et[1] * L.q[2] + R.et[2] * L.q[3];
        const double h = (L.q[0] + R.q[0]) * 0.50;
        const Recip by_hsq(L.hs + R.hs);
        r.u = by_hsq.div(fdiv(r.hunr, L.hs) + fdiv(r.hunl, R.hs));
        r.v = by_hsq.div(fdiv(r.hutr, L.hs) + fdiv(r.hutl, R.hs));
        r.a = dsqrt(g * h);
        return r;
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &par, double (&s)[3]) {
        const double dy = IXY == 1 ? par.v[2] : par.v[1];
        const Roe r = roe(L, R, par.v[0]);
        const Recip by_dy(dy);
        s[0] = by_dy.div((r.u - r.a) * R.gam);
        s[1] = by_dy.div(r.u * R.gam);
        s[2] = by_dy.div((r.u + r.a) * R.gam);
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &par,
                                                 double (&wave)[3][4], double (&s)[3], double (&amdq)[4],
                                                 double (&apdq)[4]) {
        const double g = par.v[0];
        const double dy = IXY == 1 ? par.v[2] : par.v[1];
        const Recip by_dy(dy);
        const Roe r = roe(L, R, g);
        const double u = r.u, v = r.v, a = r.a, gamma = R.gam;
        const double delta1 = R.q[0] - L.q[0];
        const double delta2 = r.hunl - r.hunr;
        const double delta3 = r.hutl - r.hutr;
        const double hba = fdiv(0.50, a);
        const double a1 = ((u + a) * delta1 - delta2) * hba;
        const double a2 = -v * delta1 + delta3;
        const double a3 = (-(u - a) * delta1 + delta2) * hba;
        wave[0][0] = a1;
        wave[1][0] = 0.0;
        wave[2][0] = a3;
        for (int k = 0; k < 3; k++) {
            wave[0][1 + k] = a1 * (u - a) * R.en[k] + a1 * v * R.et[k];
            wave[1][1 + k] = a2 * R.et[k];
            wave[2][1 + k] = a3 * (u + a) * R.en[k] + a3 * v * R.et[k];
        }
        s[0] = by_dy.div((u - a) * gamma);
        s[1] = by_dy.div(u * gamma);
        s[2] = by_dy.div((u + a) * gamma);

        // Harten-Hyman entropy fix; the Fortran's early exits become flags
        for (int m = 0; m < 4; m++) amdq[m] = 0.0;
        const double s0 = by_dy.div((fdiv(r.hunr, L.q[0]) - L.sgh) * gamma);
        bool done = (s0 > 0.0) && (s[0] > 0.0);
        {
            const double h1 = L.q[0] + wave[0][0];
            const double hu1 = r.hunr + R.en[0] * wave[0][1] + R.en[1] * wave[0][2] + R.en[2] * wave[0][3];
            const double s1 = by_dy.div((fdiv(hu1, h1) - dsqrt(g * h1)) * gamma);
            double sfract;
            if (s0 < 0.0 && s1 > 0.0) sfract = s0 * fdiv(s1 - s[0], s1 - s0);
            else if (s[0] < 0.0) sfract = s[0];
            else sfract = 0.0;
            if (!done)
                for (int m = 0; m < 4; m++) amdq[m] = sfract * wave[0][m];
        }
        done = done || (s[1] > 0.0);
        {
            const double s03 = by_dy.div((fdiv(r.hunl, R.q[0]) + R.sgh) * gamma);
            const double h3 = R.q[0] - wave[2][0];
            const double hu3 = r.hunl - (R.en[0] * wave[2][1] + R.en[1] * wave[2][2] + R.en[2] * wave[2][3]);
            const double s3 = by_dy.div((fdiv(hu3, h3) + dsqrt(g * h3)) * gamma);
            double sfract = 0.0;
            bool add = true;
            if (s3 < 0.0 && s03 > 0.0) sfract = s3 * fdiv(s03 - s[2], s03 - s3);
            else if (s[2] < 0.0) sfract = s[2];
            else add = false;
            if (!done) {
                for (int m = 1; m < 4; m++) amdq[m] = amdq[m] + s[1] * wave[1][m];
                if (add)
                    for (int m = 0; m < 4; m++) amdq[m] = amdq[m] + sfract * wave[2][m];
            }
        }
        for (int m = 0; m < 4; m++) {
            double df = s[0] * wave[0][m];
            if (m > 0) df = df + s[1] * wave[1][m];
            df = df + s[2] * wave[2][m];
            apdq[m] = df - amdq[m];
        }
        // project the momentum parts onto the tangent plane of the cell each fluctuation enters
        const double amn = L.er[0] * amdq[1] + L.er[1] * amdq[2] + L.er[2] * amdq[3];
        const double apn = R.er[0] * apdq[1] + R.er[1] * apdq[2] + R.er[2] * apdq[3];
        for (int k = 0; k < 3; k++) {
            amdq[1 + k] = amdq[1 + k] - amn * L.er[k];
            apdq[1 + k] = apdq[1 + k] - apn * R.er[k];
        }
    }
    // asdq sits in cell c1; auxo / auxb / auxa = (other-direction edge normal 0-2, tangent 3-5, radial vector 6-8) of
    // that cell and of its neighbours below / above.  Up-going: the edge ABOVE the cell (the neighbour's own edge),
    // tangent plane of the cell above; down-going: the cell's own edge, tangent plane of the cell below.
    template <int IXY>
    __device__ static __forceinline__ void transverse_vc(const Cell &c1, const double *auxo, const double *auxb,
                                                         const double *auxa, const RpParams &par,
                                                         const double (&asdq)[4], double (&bm)[4], double (&bp)[4]) {
        const double dx = IXY == 1 ? par.v[1] : par.v[2];
        const Recip by_dx(dx);
        const double h = c1.q[0];
        const double a = c1.sgh;
        const double hba = fdiv(0.50, a);
        const Recip by_h(h);
#pragma unroll
        for (int up = 1; up >= 0; up--) {
            const double *ae = up ? auxa : auxo;
            const double *ap = up ? auxa : auxb;
            const double enx = ae[0], eny = ae[1], enz = ae[2];
            double etx = ae[3], ety = ae[4], etz = ae[5];
            const double gamma = dsqrt(etx * etx + ety * ety + etz * etz);
            const Recip by_gam(gamma);
            etx = by_gam.div(etx); ety = by_gam.div(ety); etz = by_gam.div(etz);
            const double u = by_h.div(enx * c1.q[1] + eny * c1.q[2] + enz * c1.q[3]);
            const double v = by_h.div(etx * c1.q[1] + ety * c1.q[2] + etz * c1.q[3]);
            const double delta1 = asdq[0];
            const double delta2 = enx * asdq[1] + eny * asdq[2] + enz * asdq[3];
            const double delta3 = etx * asdq[1] + ety * asdq[2] + etz * asdq[3];
            const double a1 = ((u + a) * delta1 - delta2) * hba;
            const double a2 = -v * delta1 + delta3;
            const double a3 = (-(u - a) * delta1 + delta2) * hba;
            const double en[3] = {enx, eny, enz}, et[3] = {etx, ety, etz};
            double wb[3][4], sb[3];
            wb[0][0] = a1; wb[1][0] = 0.0; wb[2][0] = a3;
            for (int k = 0; k < 3; k++) {
                wb[0][1 + k] = a1 * (u - a) * en[k] + a1 * v * et[k];
                wb[1][1 + k] = a2 * et[k];
                wb[2][1 + k] = a3 * (u + a) * en[k] + a3 * v * et[k];
            }
            sb[0] = by_dx.div((u - a) * gamma);
            sb[1] = by_dx.div(u * gamma);
            sb[2] = by_dx.div((u + a) * gamma);
            double out[4];
            for (int m = 0; m < 4; m++) {
                double acc = 0.0;
                for (int mw = 0; mw < 3; mw++)
                    acc = acc + (up ? dmax(sb[mw], 0.0) : dmin(sb[mw], 0.0)) * wb[mw][m];
                out[m] = acc;
            }
            const double bn = ap[6] * out[1] + ap[7] * out[2] + ap[8] * out[3];
            for (int k = 0; k < 3; k++) out[1 + k] = out[1 + k] - bn * ap[6 + k];
            for (int m = 0; m < 4; m++) {
                if (up) bp[m] = out[m];
                else bm[m] = out[m];
            }
        }
    }
    // qcor.f:1-72: el / er = raw (normal 0-2, tangent 3-5) of the cell's own edge and of the next cell's edge along the
    // sweep, rad = the cell's radial vector
    template <int IXY>
    __device__ static __forceinline__ void qcor(const double *q, const double *el, const double *er_, const double *rad,
                                                const RpParams &par, double (&qc)[4]) {
        const double g = par.v[0];
        const double dy = IXY == 1 ? par.v[2] : par.v[1];
        const double gammal = fdiv_ieee(dsqrt(el[3] * el[3] + el[4] * el[4] + el[5] * el[5]), dy);
        const double enxl = el[0] * gammal, enyl = el[1] * gammal, enzl = el[2] * gammal;
        const double gammar = fdiv_ieee(dsqrt(er_[3] * er_[3] + er_[4] * er_[4] + er_[5] * er_[5]), dy);
        const double enxr = er_[0] * gammar, enyr = er_[1] * gammar, enzr = er_[2] * gammar;
        const double q1 = q[0], q2 = q[1], q3 = q[2], q4 = q[3];
        const Recip by_q1(q1);
        const double hg = 0.5 * g * (q1 * q1);
        const double dx_ = enxr - enxl, dy_ = enyr - enyl, dz_ = enzr - enzl;
        qc[0] = dx_ * q2 + dy_ * q3 + dz_ * q4;
        qc[1] = dx_ * (by_q1.div(q2 * q2) + hg) + dy_ * by_q1.div(q2 * q3) + dz_ * by_q1.div(q2 * q4);
        qc[2] = dx_ * by_q1.div(q2 * q3) + dy_ * (by_q1.div(q3 * q3) + hg) + dz_ * by_q1.div(q3 * q4);
        qc[3] = dx_ * by_q1.div(q2 * q4) + dy_ * by_q1.div(q3 * q4) + dz_ * (by_q1.div(q4 * q4) + hg);
        const double qcn = rad[0] * qc[1] + rad[1] * qc[2] + rad[2] * qc[3];
        for (int k = 0; k < 3; k++) qc[1 + k] = qc[1 + k] - qcn * rad[k];
    }
};

// ------------------------------------------------------------------------------------
// 2-D Euler, Roe solver with 5 waves (acoustic-, shear, entropy, acoustic+, tracer) and
// the Harten-Hyman entropy fix: development/rp_approaches/rpn2_euler_5wave.f:87-298,
// transverse split rpt2_euler_5wave_rec_loc.f:50-116.  par = gamma, gamma1.
// ------------------------------------------------------------------------------------
struct Euler5 {
    static constexpr int MEQN = 5, MWAVES = 5, NCELL = 12, NAUX = 0;
    struct Cell {
        double q[5];
        double rs;      // sqrt(rho)                       rpn2:88-89
        double p;       // pressure                        rpn2:90-93 == :211-212,260-261
        double qu_rs;   // q(mu)/sqrt(rho)                 rpn2:95
        double qv_rs;   // q(mv)/sqrt(rho)                 rpn2:96
        double h_rs;    // (E+p)/sqrt(rho)                 rpn2:97-98
        double smc;     // q(mu)/rho - sqrt(gamma*p/rho)   rpn2:213-215  (u - c of this cell: the 1-wave check on its right edge)
        double spc;     // q(mu)/rho + sqrt(gamma*p/rho)   rpn2:262-264  (u + c: the 3-wave check on its left edge)
    };
    // wave(m,mw) sparsity, rpn2:124-163
    template <int IXY> __device__ static constexpr bool nz(int mw, int m) {
        constexpr int mv = (IXY == 1) ? 2 : 1;
        return mw == 1 ? (m == mv || m == 3) : mw == 4 ? (m == 4) : (m != 4);
    }
    template <int IXY>
    __device__ static __forceinline__ Cell precell(const double *q, const RpParams &par) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double gamma = par.v[0], gamma1 = par.v[1];
        Cell c;
        for (int m = 0; m < 5; m++) c.q[m] = q[m];
        const SqrtH sr = dsqrt_pos_h(q[0]);
        c.rs = sr.g;
        // 1/sqrt(rho) seeded with 2h, 1/rho with its square: same quotients as Recip(q[0]), Recip(c.rs)
        const Recip by_rs = Recip::seeded(c.rs, sr.h + sr.h);
        const Recip by_rho = Recip::seeded(q[0], by_rs.r * by_rs.r);
        c.p = gamma1 * (q[3] - by_rho.div(0.5 * (q[1] * q[1] + q[2] * q[2])));
        c.qu_rs = by_rs.div(q[mu]);
        c.qv_rs = by_rs.div(q[mv]);
        c.h_rs = by_rs.div(q[3] + c.p);
        const double cc = dsqrt(by_rho.div(gamma * c.p)), un = by_rho.div(q[mu]);
        c.smc = un - cc;
        c.spc = un + cc;
        return c;
    }
    struct Roe { double u, v, enth, a, g1a2, euv, u2v2; Recip by_2a; };
    __device__ static __forceinline__ Roe roe(const Cell &L, const Cell &R, double gamma1) {
        Roe r;
        const Recip by_rhsq2(L.rs + R.rs);
        r.u = by_rhsq2.div(L.qu_rs + R.qu_rs);
        r.v = by_rhsq2.div(L.qv_rs + R.qv_rs);
        r.enth = by_rhsq2.div(L.h_rs + R.h_rs);
        r.u2v2 = r.u * r.u + r.v * r.v;
        const double a2 = gamma1 * (r.enth - .5 * r.u2v2);
        const SqrtH sa = dsqrt_pos_h(a2);
        r.a = sa.g;
        // 1/(2a) seeded with h (the a4 / transverse a4 quotients), 1/a^2 with 4*(1/(2a))^2
        r.by_2a = Recip::seeded(2.0 * r.a, sa.h);
        r.g1a2 = Recip::seeded(a2, 4.0 * (r.by_2a.r * r.by_2a.r)).div(gamma1);
        r.euv = r.enth - r.u2v2;
        return r;
    }
    template <int IXY>
    __device__ static __forceinline__ void speeds(const Cell &L, const Cell &R, const RpParams &par, double (&s)[5]) {
        const Roe r = roe(L, R, par.v[1]);
        s[0] = r.u - r.a; s[1] = r.u; s[2] = r.u; s[3] = r.u + r.a; s[4] = r.u;
    }
    template <int IXY>
    __device__ static __forceinline__ void solve(const Cell &L, const Cell &R, const RpParams &par,
                                                 double (&wave)[5][5], double (&s)[5],
                                                 double (&amdq)[5], double (&apdq)[5]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double gamma = par.v[0], gamma1 = par.v[1];
        const Roe r = roe(L, R, gamma1);
        const double u = r.u, v = r.v, enth = r.enth, a = r.a;
        const double delta1 = R.q[0] - L.q[0];
        const double delta2 = R.q[mu] - L.q[mu];
        const double delta3 = R.q[mv] - L.q[mv];
        const double delta4 = R.q[3] - L.q[3];
        const double a3 = r.g1a2 * (r.euv * delta1 + u * delta2 + v * delta3 - delta4);
        const double a2 = delta3 - v * delta1;
        const double a4 = r.by_2a.div(delta2 + (a - u) * delta1 - a * a3);
        const double a1 = delta1 - a3 - a4;

        wave[0][0] = a1; wave[0][mu] = a1 * (u - a); wave[0][mv] = a1 * v;
        wave[0][3] = a1 * (enth - u * a); wave[0][4] = 0.0; s[0] = u - a;
        wave[1][0] = 0.0; wave[1][mu] = 0.0; wave[1][mv] = a2;
        wave[1][3] = a2 * v; wave[1][4] = 0.0; s[1] = u;
        wave[2][0] = a3; wave[2][mu] = a3 * u; wave[2][mv] = a3 * v;
        wave[2][3] = a3 * 0.5 * r.u2v2; wave[2][4] = 0.0; s[2] = u;
        wave[3][0] = a4; wave[3][mu] = a4 * (u + a); wave[3][mv] = a4 * v;
        wave[3][3] = a4 * (enth + u * a); wave[3][4] = 0.0; s[3] = u + a;
        wave[4][0] = 0.0; wave[4][mu] = 0.0; wave[4][mv] = 0.0; wave[4][3] = 0.0;
        wave[4][4] = R.q[4] - L.q[4]; s[4] = u;

        // ---- entropy fix (rpn2:205-286).  The early exits of the Fortran become flags.
        const double s0 = L.smc;                            // u-c in left state
        const bool all_right = (s0 >= 0.0) && (s[0] > 0.0); // rpn2:217
        {
            const double rho1 = L.q[0] + wave[0][0];
            const double rhou1 = L.q[mu] + wave[0][mu];
            const double rhov1 = L.q[mv] + wave[0][mv];
            const double en1 = L.q[3] + wave[0][3];
            const Recip by_rho1(rho1);
            const double p1 = gamma1 * (en1 - by_rho1.div(0.5 * (rhou1 * rhou1 + rhov1 * rhov1)));
            const double c1 = dsqrt(by_rho1.div(gamma * p1));
            const double s1 = by_rho1.div(rhou1) - c1;
            double sfract;
            if (s0 < 0.0 && s1 > 0.0)
                sfract = fdiv(s0 * (s1 - s[0]), s1 - s0);
            else if (s[0] < 0.0)
                sfract = s[0];
            else
                sfract = 0.0;
            for (int m = 0; m < 5; m++) amdq[m] = sfract * wave[0][m];
        }
        if (!(s[1] >= 0.0)) {                                // rpn2:249
            for (int m = 0; m < 5; m++) {
                if (nz<IXY>(1, m)) amdq[m] = amdq[m] + s[1] * wave[1][m];
                if (nz<IXY>(2, m)) amdq[m] = amdq[m] + s[2] * wave[2][m];
                if (nz<IXY>(4, m)) amdq[m] = amdq[m] + s[4] * wave[4][m];
            }
            const double s3 = R.spc;                         // u+c in right state
            const double rho2 = R.q[0] - wave[3][0];
            const double rhou2 = R.q[mu] - wave[3][mu];
            const double rhov2 = R.q[mv] - wave[3][mv];
            const double en2 = R.q[3] - wave[3][3];
            const Recip by_rho2(rho2);
            const double p2 = gamma1 * (en2 - by_rho2.div(0.5 * (rhou2 * rhou2 + rhov2 * rhov2)));
            const double c2 = dsqrt(by_rho2.div(gamma * p2));
            const double s2 = by_rho2.div(rhou2) + c2;
            double sfract = 0.0;
            bool add4 = true;
            if (s2 < 0.0 && s3 > 0.0)
                sfract = fdiv(s2 * (s3 - s[3]), s3 - s2);
            else if (s[3] < 0.0)
                sfract = s[3];
            else
                add4 = false;
            if (add4)
                for (int m = 0; m < 4; m++) amdq[m] = amdq[m] + sfract * wave[3][m];
        }
        if (all_right)
            for (int m = 0; m < 5; m++) amdq[m] = 0.0;

        // apdq = sum_mw s*wave - amdq  (rpn2:291-298), zero wave entries skipped
        for (int m = 0; m < 5; m++) {
            double df = 0.0;
            bool first = true;
            for (int mw = 0; mw < 5; mw++)
                if (nz<IXY>(mw, m)) {
                    df = first ? s[mw] * wave[mw][m] : df + s[mw] * wave[mw][m];
                    first = false;
                }
            apdq[m] = df - amdq[m];
        }
    }
    template <int IXY>
    __device__ static __forceinline__ void transverse(const Cell &L, const Cell &R,
                                                      const RpParams &par, const double (&asdq)[5],
                                                      double (&bm)[5], double (&bp)[5]) {
        constexpr int mu = (IXY == 1) ? 1 : 2, mv = (IXY == 1) ? 2 : 1;
        const double gamma1 = par.v[1];
        const Roe r = roe(L, R, gamma1);
        const double u = r.u, v = r.v, enth = r.enth, a = r.a;
        const double a3 = r.g1a2 * (r.euv * asdq[0] + u * asdq[mu] + v * asdq[mv] - asdq[3]);
        const double a2 = asdq[mu] - u * asdq[0];
        const double a4 = r.by_2a.div(asdq[mv] + (a - v) * asdq[0] - a * a3);
        const double a1 = asdq[0] - a3 - a4;
        double wb[4][5], sb[4];
        wb[0][0] = a1; wb[0][mu] = a1 * u; wb[0][mv] = a1 * (v - a);
        wb[0][3] = a1 * (enth - v * a); wb[0][4] = 0.0; sb[0] = v - a;
        wb[1][0] = a3; wb[1][mu] = a3 * u + a2; wb[1][mv] = a3 * v;
        wb[1][3] = a3 * 0.5 * r.u2v2 + a2 * u; wb[1][4] = 0.0; sb[1] = v;
        wb[2][0] = a4; wb[2][mu] = a4 * u; wb[2][mv] = a4 * (v + a);
        wb[2][3] = a4 * (enth + v * a); wb[2][4] = 0.0; sb[2] = v + a;
        wb[3][0] = 0.0; wb[3][mu] = 0.0; wb[3][mv] = 0.0; wb[3][3] = 0.0;
        wb[3][4] = asdq[4]; sb[3] = v;
        for (int m = 0; m < 5; m++) {
            double m_ = 0.0, p_ = 0.0;
            for (int mw = 0; mw < 4; mw++) {
                m_ = m_ + dmin(sb[mw], 0.0) * wb[mw][m];
                p_ = p_ + dmax(sb[mw], 0.0) * wb[mw][m];
            }
            bm[m] = m_; bp[m] = p_;
        }
    }
};

}  // namespace PCL_NS
}  // namespace pcl
