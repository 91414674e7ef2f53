// sharpclaw.hpp -- SharpClaw semi-discrete right-hand side (dq = dt * dq/dt) as a gfx950 kernel.
//
// Reference path restated (operation order preserved):
//   flux2 (slice driver)   src/fortran/2d/sharpclaw/flux2.f90:32-94
//   flux1                  src/fortran/2d/sharpclaw/flux1.f90:59-188   (char_decomp=0, no tfluct)
//   weno5, PyWENO form     src/fortran/1d/sharpclaw/weno.f90:36-100    (lim_type=2)
//   weno5, legacy form     src/fortran/1d/sharpclaw/reconstruct.f90:120-185 (lim_type=3)
//   tvd2                   src/fortran/1d/sharpclaw/reconstruct.f90:568-625 (lim_type=1)
//
// Same mapping as the classic kernel (classic.hpp): one lane = one cell of a 64-cell strip, a
// 64 x 16 tile staged through LDS, but a 3-cell halo (WENO5 needs q(i-2..i+2) and the update needs
// the interface on either side): lanes 3..60 produce dq.  Per cell: one WENO reconstruction
// (its stencil read straight from the LDS tile), the Riemann problem at interface i (left
// neighbour's right-edge state arrives by wavefront shift) and the Riemann problem INSIDE the
// cell between its own edge states (flux1.f90:166-187).
#pragma once
#include "classic.hpp"
#include "weno_tables.hpp"

namespace pcl {
namespace PCL_NS {

constexpr int T_ACROSS_S = 16;
// strips per tile: 16 (a 128-byte row segment per tile row) except in the y pass of solvers that stage many aux planes
// through the tile (the sphere solver: 4 + 1 + 9 planes): 8 columns = 57 KB instead of 114 KB, two workgroups per CU
template <class RP, int IXY> constexpr int sharp_tile_across() { return (IXY == 2 && RP::NAUX >= 6) ? 8 : T_ACROSS_S; }
// halo of a strip = mbc = (weno_order+1)/2 (sharpclaw.py:479): 3 for WENO5 / tvd2, up to 9 for WENO17
constexpr int sstrip(int halo) { return WAVE - 2 * halo; }   // cells updated per strip: 58 ... 46

// REAL*4 literals of the generated Fortran, promoted to double (weno.f90 has no d0 exponents)
#define PCL_F32(x) ((double)(float)(x))

// weno.f90:36-100 for one component: q(i-2..i+2) -> ql(i), qr(i)
__device__ __forceinline__ void weno5_pyweno(double qm2, double qm1, double q0, double qp1, double qp2,
                                             double &ql, double &qr) {
    constexpr double c333 = PCL_F32(+3.33333333333333), c1033 = PCL_F32(-10.3333333333333);
    constexpr double c366 = PCL_F32(+3.66666666666667), c833 = PCL_F32(+8.33333333333333);
    constexpr double c633 = PCL_F32(-6.33333333333333), c133 = PCL_F32(+1.33333333333333);
    constexpr double c433m = PCL_F32(-4.33333333333333), c166 = PCL_F32(+1.66666666666667);
    constexpr double c433 = PCL_F32(+4.33333333333333);
    constexpr double eps = PCL_F32(1.0e-36), w01 = PCL_F32(+0.1), w06 = PCL_F32(+0.6), w03 = PCL_F32(+0.3);
    constexpr double f183 = PCL_F32(+1.83333333333333), f116 = PCL_F32(-1.16666666666667);
    constexpr double f033 = PCL_F32(+0.333333333333333), f083 = PCL_F32(+0.833333333333333);
    constexpr double f016 = PCL_F32(-0.166666666666667);
    const double sigma0 = ((c333 * q0) * q0) + ((c1033 * q0) * qp1) + ((c366 * q0) * qp2) +
                          ((c833 * qp1) * qp1) + ((c633 * qp1) * qp2) + ((c133 * qp2) * qp2);
    const double sigma1 = ((c133 * qm1) * qm1) + ((c433m * qm1) * q0) + ((c166 * qm1) * qp1) +
                          ((c433 * q0) * q0) + ((c433m * q0) * qp1) + ((c133 * qp1) * qp1);
    const double sigma2 = ((c133 * qm2) * qm2) + ((c633 * qm2) * qm1) + ((c366 * qm2) * q0) +
                          ((c833 * qm1) * qm1) + ((c1033 * qm1) * q0) + ((c333 * q0) * q0);
    const double t0 = sigma0 + eps, t1 = sigma1 + eps, t2 = sigma2 + eps;
    const double d0 = t0 * t0, d1 = t1 * t1, d2 = t2 * t2;
#if PCL_FAST
    // fast mode (rtol 1e-12, not bit-identical): the weights w_k / d_k, normalised, are (w_k * prod_{j != k} d_j) / sum --
    // two reciprocals instead of five and no quotient per weight; the candidates are combined BEFORE the one division.
    // Range: d >= 1e-72 and sigma ~ q^2, so the pair products stay far inside the double range (|q| < 1e38).
    {
        const double d12 = d1 * d2, d02 = d0 * d2, d01 = d0 * d1;
        const double a0 = w01 * d12, a1 = w06 * d02, a2 = w03 * d01;      // left weights  (0.1, 0.6, 0.3)
        const double b0 = w03 * d12, b2 = w01 * d01;                      // right weights (0.3, 0.6, 0.1): b1 == a1
        const double fr0 = f183 * q0 + f116 * qp1 + f033 * qp2;
        const double fr1 = f033 * qm1 + f083 * q0 + f016 * qp1;
        const double fr2 = f016 * qm2 + f083 * qm1 + f033 * q0;
        const double fr3 = f033 * q0 + f083 * qp1 + f016 * qp2;
        const double fr4 = f016 * qm1 + f083 * q0 + f033 * qp1;
        const double fr5 = f033 * qm2 + f116 * qm1 + f183 * q0;
        ql = (a0 * fr0 + a1 * fr1 + a2 * fr2) * Recip(a0 + a1 + a2).r;
        qr = (b0 * fr3 + a1 * fr4 + b2 * fr5) * Recip(b0 + a1 + b2).r;
        return;
    }
#endif
    // 12 divisions, 5 distinct denominators (each (sigma+eps)^2 >= 1e-72 is used twice, each weight
    // sum three times): shared-reciprocal quotients, the same correctly rounded values (rp.hpp).
    const Recip by_d0(d0), by_d1(d1), by_d2(d2);
    double acc = 0.0;
    double omega0 = by_d0.div(w01); acc = acc + omega0;
    double omega1 = by_d1.div(w06); acc = acc + omega1;
    double omega2 = by_d2.div(w03); acc = acc + omega2;
    const Recip by_acc(acc);
    omega0 = by_acc.div(omega0); omega1 = by_acc.div(omega1); omega2 = by_acc.div(omega2);
    acc = 0.0;
    double omega3 = by_d0.div(w03); acc = acc + omega3;
    double omega4 = by_d1.div(w06); acc = acc + omega4;
    double omega5 = by_d2.div(w01); acc = acc + omega5;
    const Recip by_acc2(acc);
    omega3 = by_acc2.div(omega3); omega4 = by_acc2.div(omega4); omega5 = by_acc2.div(omega5);
    const double fr0 = f183 * q0 + f116 * qp1 + f033 * qp2;
    const double fr1 = f033 * qm1 + f083 * q0 + f016 * qp1;
    const double fr2 = f016 * qm2 + f083 * qm1 + f033 * q0;
    const double fr3 = f033 * q0 + f083 * qp1 + f016 * qp2;
    const double fr4 = f016 * qm1 + f083 * q0 + f033 * qp1;
    const double fr5 = f033 * qm2 + f116 * qm1 + f183 * q0;
    ql = omega0 * fr0 + omega1 * fr1 + omega2 * fr2;
    qr = omega3 * fr3 + omega4 * fr4 + omega5 * fr5;
}

// weno.f90: weno7 ... weno17 (K = 4..9; K = 3 is the hand-unrolled weno5 above): for the K stencils r = 0..K-1 (cells
// i-r .. i-r+K-1) the smoothness indicator as a sum over a <= b of ((C*q_a)*q_b) in lexicographic order, the two sets
// of nonlinear weights w / (sigma + 1e-36)**2 normalised by their running sum, the 2K candidate values and their
// weighted sums, every sum left-associated in the printed order.  Tables: weno_tables.hpp (tools/gen_weno.py).
// qs[j] = q(i - (K-1) + j).  The 2K + 2K divisions share K + 2 reciprocals (same correctly rounded quotients).
template <int K>
__device__ __forceinline__ void weno_pyweno_k(const double (&qs)[2 * K - 1], double &ql, double &qr) {
    // (Conditioning: the coefficients of the high orders reach 1e5 with alternating signs, so on smooth data the
    // smoothness sums are rounding noise and the weights -- w / (sigma + 1e-36)^2 -- follow that noise.  Bit-identical
    // to the reference in exact mode; in fast mode, or for inputs that differ in the last digit, the stencils are
    // weighted differently: another valid WENO result, not a nearby one.  See DESIGN 4.3.)
    constexpr int T = K - 3;
    constexpr double eps = PCL_F32(1.0e-36);
    Recip by_d[K];
#pragma unroll
    for (int r = 0; r < K; r++) {
        double sg = 0.0;
#pragma unroll
        for (int a = 0; a < K; a++)
#pragma unroll
            for (int b = a; b < K; b++) {
                const double term = (WENO_SIG[T][r][a][b] * qs[K - 1 - r + a]) * qs[K - 1 - r + b];
                sg = (a == 0 && b == 0) ? term : sg + term;
            }
        const double t = sg + eps;
        by_d[r] = Recip(t * t);
    }
    double oml[K], omr[K], accl = 0.0, accr = 0.0;
#pragma unroll
    for (int r = 0; r < K; r++) {
        oml[r] = by_d[r].div(WENO_WL[T][r]);
        omr[r] = by_d[r].div(WENO_WR[T][r]);
        accl = r == 0 ? oml[r] : accl + oml[r];
        accr = r == 0 ? omr[r] : accr + omr[r];
    }
    const Recip by_l(accl), by_r(accr);
    double fs0 = 0.0, fs1 = 0.0;
#pragma unroll
    for (int r = 0; r < K; r++) {
        double fl = 0.0, fr = 0.0;
#pragma unroll
        for (int j = 0; j < K; j++) {
            const double tl = WENO_CL[T][r][j] * qs[K - 1 - r + j], tr = WENO_CR[T][r][j] * qs[K - 1 - r + j];
            fl = j == 0 ? tl : fl + tl;
            fr = j == 0 ? tr : fr + tr;
        }
        const double wl = by_l.div(oml[r]), wr = by_r.div(omr[r]);
        fs0 = r == 0 ? wl * fl : fs0 + wl * fl;
        fs1 = r == 0 ? wr * fr : fs1 + wr * fr;
    }
    ql = fs0;
    qr = fs1;
}

// reconstruct.f90:147-176: the interface value uu(m1,i) between cells i-1 and i.
// d2,d1,d0,dp1 = dq1m(i+intwo), dq1m(i+inone), dq1m(i), dq1m(i+ione); im = +1 (from the left) / -1
__device__ __forceinline__ double weno5_legacy_edge(double im, double d2, double d1, double d0, double dp1,
                                                    double qim2, double qim1, double qi, double qip1) {
    constexpr double epweno = PCL_F32(1.e-36);
    const double t1 = im * (d2 - d1);
    const double t2 = im * (d1 - d0);
    const double t3 = im * (d0 - dp1);
    const double a1 = d2 - 3. * d1, a2 = d1 + d0, a3 = 3. * d0 - dp1;
    double tt1 = 13. * (t1 * t1) + 3. * (a1 * a1);
    double tt2 = 13. * (t2 * t2) + 3. * (a2 * a2);
    double tt3 = 13. * (t3 * t3) + 3. * (a3 * a3);
    tt1 = (epweno + tt1) * (epweno + tt1);
    tt2 = (epweno + tt2) * (epweno + tt2);
    tt3 = (epweno + tt3) * (epweno + tt3);
    double s1 = tt2 * tt3;
    const double s2 = 6. * tt1 * tt3;
    double s3 = 3. * tt1 * tt2;
    const double t0 = fdiv_ieee(1., s1 + s2 + s3);
    s1 = s1 * t0;
    s3 = s3 * t0;
    return (s1 * (t2 - t1) + (0.5 * s3 - 0.25) * (t3 - t2)) / 3. + (-qim2 + 7. * (qim1 + qi) - qip1) / 12.;
}

// legacy weno5 for one component of cell c: ql(c) = uu(2,c), qr(c) = uu(1,c+1)
__device__ __forceinline__ void weno5_legacy(double qm2, double qm1, double q0, double qp1, double qp2,
                                             double &ql, double &qr) {
    const double dm1 = qm1 - qm2;  // dq1m(c-1)
    const double d0 = q0 - qm1;    // dq1m(c)
    const double dp1 = qp1 - q0;   // dq1m(c+1)
    const double dp2 = qp2 - qp1;  // dq1m(c+2)
    // uu(2,i=c): im=-1, ione=-1, inone=+1, intwo=+2; central part uses q(i-2..i+1) = q(c-2..c+1)
    ql = weno5_legacy_edge(-1.0, dp2, dp1, d0, dm1, qm2, qm1, q0, qp1);
    // uu(1,i=c+1): im=+1, intwo=-2 -> dq1m(c-1), inone=-1 -> dq1m(c), dq1m(c+1), ione -> dq1m(c+2);
    // central part q(i-2..i+1) = q(c-1..c+2)
    qr = weno5_legacy_edge(+1.0, dm1, d0, dp1, dp2, qm1, q0, qp1, qp2);
}

// reconstruct.f90:568-625 (tvd2) for one component of cell c: dqm = q(c) - q(c-1), dqp = q(c+1) - q(c);
// meth = mthlim(m) -- indexed by COMPONENT in the reference.  (The Fortran reads an uninitialised dqm at the first
// cell of each slice: ghost cell 0 only; see oracle/sharpclaw_oracle.c.  0/0 on constant data: v_max/min_f64 drop
// the NaN like gfortran's MAX/MIN, ql = qr = q.)
__device__ __forceinline__ void tvd2_cell(double qm1, double q0, double qp1, int meth, double &ql, double &qr) {
    const double dqm = q0 - qm1, dqp = qp1 - q0;
    const double r = fdiv_ieee(dqp, dqm);
    double lim = 0.0;
    switch (meth) {
    case 1: lim = dmax(0.0, dmin(1.0, r)); break;
    case 2: lim = dmax(dmax(0.0, dmin(1.0, 2.0 * r)), dmin(2.0, r)); break;
    case 3: lim = fdiv_ieee(r + fabs(r), 1.0 + fabs(r)); break;
    case 4: { const double c = (1.0 + r) / 2.0; lim = dmax(0.0, dmin(dmin(c, 2.0), 2.0 * r)); break; }
    case 5: {
        const double alpha = 1.0 / 3.0;
        const double pp = fdiv_ieee(2.0 + r, 3.0);
        const double amax = dmax(dmax(-alpha * r, 0.0), dmin(dmin(2.0 * r, pp), 2.0));
        lim = dmax(0.0, dmin(pp, amax));
        break;
    }
    }
    qr = q0 + 0.5 * lim * dqm;
    ql = q0 - 0.5 * lim * dqm;
}

// ---- the kernel ----------------------------------------------------------------------------------
// x pass (IXY=1): dq(interior) = dq1d ; y pass (IXY=2): dq += dq1d   (flux2.f90:54-56,86-88)
// y pass: rows of 16 doubles with the column XOR-swizzled by (al >> 1): a wavefront reading one column (lanes =
// rows) touches 32 distinct 8-byte banks per half-wave, a thread group reading a row touches 16 consecutive ones
// -- conflict-free both ways without the 17th padding column, which keeps the tile at 40 KB (4 per CU).
template <int IXY, int TA = T_ACROSS_S> __device__ __forceinline__ int stile_at(int m, int al, int ac) {
    return IXY == 1 ? (m * TA + ac) * WAVE + al : (m * WAVE + al) * TA + (ac ^ ((al >> 1) & (TA - 1)));
}

#ifndef PCL_SHARP_OCC
#define PCL_SHARP_OCC 4     // workgroups per CU the register budget is sized for (A/B: build with -DPCL_SHARP_OCC=3)
#endif
template <class RP, int IXY, bool CAPA, int LIM, int K = 3, bool SRC = false>
__global__ __launch_bounds__(256, CAPA ? 3 : PCL_SHARP_OCC) void sharp_kernel(SweepArgs a, int ntiles_across, int ntiles_along) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES;
    constexpr int SHALO = K, SSTRIP = sstrip(K);      // a.mbc == K (checked by the launcher)
    static_assert(K == 3 || LIM == 2, "orders above 5 exist for the PyWENO form only");
    constexpr int NAUX = RP::NAUX, PAUX = MEQN + (CAPA ? 1 : 0);   // planes: q, capa, the RP's aux components
    // x pass: the lanes of a strip run along i, so each lane reads its own aux values straight from HBM (coalesced)
    // and the tile holds only q (+ capa): 40-48 KB instead of 114 KB for the sphere solver's 9 components, i.e.
    // 3-4 workgroups per CU instead of 1.  y pass: aux goes through the tile like q (the transposition).
    constexpr int NAUX_LDS = IXY == 1 ? 0 : NAUX;
    constexpr int NP = PAUX + NAUX_LDS;
    constexpr int TA = sharp_tile_across<RP, IXY>();     // strips (columns / rows across the sweep) per tile
    constexpr int NLD = TA * WAVE / 256;                   // cells each thread loads / stores
    constexpr int PLANE = TA * WAVE;
    using Cell = typename RP::Cell;
    __shared__ double tile[NP * PLANE];

    const int n_along = IXY == 1 ? a.I : a.J;
    const int n_across = IXY == 1 ? a.J : a.I;
    const int m_along = IXY == 1 ? a.mx : a.my;
    const int m_across = IXY == 1 ? a.my : a.mx;
    const int tb = IXY == 1 ? blockIdx.x / ntiles_along : blockIdx.x % ntiles_across;
    const int ta = IXY == 1 ? blockIdx.x % ntiles_along : blockIdx.x / ntiles_across;
    if (IXY == 1 && a.sub != 0) {   // decomposed run: tiles that read no ghost cell (box) / the others (pclaw.hip)
        const bool inside = tb >= a.box[0] && tb < a.box[1] && ta >= a.box[2] && ta < a.box[3];
        if ((a.sub == 1) != inside) return;             // workgroup-uniform, before any barrier
    }
    const int b0 = IXY == 1 ? tb * TA : tb * TA - (LINE - a.mbc);
    const int a0 = a.mbc - SHALO + ta * SSTRIP;

    // cooperative load (memory-contiguous index fastest)
    const int l_al = IXY == 1 ? threadIdx.x % WAVE : threadIdx.x / TA;
    const int l_ac = IXY == 1 ? threadIdx.x / WAVE : threadIdx.x % TA;
    constexpr int STEP_AL = IXY == 1 ? 0 : 256 / TA;
    constexpr int STEP_AC = IXY == 1 ? 256 / WAVE : 0;
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int al = l_al + k * STEP_AL, ac = l_ac + k * STEP_AC;
        int ga = a0 + al, gb = b0 + ac;
        ga = ga < n_along ? ga : n_along - 1;
        gb = gb < 0 ? 0 : (gb < n_across ? gb : n_across - 1);
        const long g = IXY == 1 ? (long)gb * a.pitch + ga : (long)ga * a.pitch + gb;
#pragma unroll
        for (int m = 0; m < MEQN; m++) tile[stile_at<IXY, TA>(m, al, ac)] = a.qin[m * a.plane + g];
        if constexpr (CAPA) tile[stile_at<IXY, TA>(MEQN, al, ac)] = a.aux[(long)(a.mcapa - 1) * a.plane + g];
#pragma unroll
        for (int m = 0; m < NAUX_LDS; m++) tile[stile_at<IXY, TA>(PAUX + m, al, ac)] = a.aux[aux_idx<RP, IXY>(m) * a.plane + g];
    }
    __syncthreads();

    const int lane = threadIdx.x & (WAVE - 1);
    const int wv = threadIdx.x / WAVE;
    const int ca = a0 + lane;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + m_along) && lane >= SHALO && lane < WAVE - SHALO;
    const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + m_along) && lane >= SHALO && lane <= WAVE - SHALO;
    const bool one_d = a.J == 1;  // 1-D grids have no transverse ghost layers
    // stencil positions clamped inside the strip (end lanes never feed a stored value)
    const int lm2 = lane >= 2 ? lane - 2 : 0, lm1 = lane >= 1 ? lane - 1 : 0;
    const int lp1 = lane <= WAVE - 2 ? lane + 1 : WAVE - 1, lp2 = lane <= WAVE - 3 ? lane + 2 : WAVE - 1;
    double cflmax = 0.0;
    for (int ac = wv; ac < TA; ac += 256 / WAVE) {
        const int gb = b0 + ac;
        if (gb >= n_across) break;  // wave-uniform
        // flux2.f90:38,70: slices 0..m+1 only (one ghost layer)
        if (!one_d && (gb < a.mbc - 1 || gb > a.mbc + m_across)) continue;
        double ql[MEQN], qr[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            const double qm2 = tile[stile_at<IXY, TA>(m, lm2, ac)], qm1 = tile[stile_at<IXY, TA>(m, lm1, ac)];
            const double q0 = tile[stile_at<IXY, TA>(m, lane, ac)];
            const double qp1 = tile[stile_at<IXY, TA>(m, lp1, ac)], qp2 = tile[stile_at<IXY, TA>(m, lp2, ac)];
            if constexpr (K > 3) {
                double qs[2 * K - 1];
#pragma unroll
                for (int j = 0; j < 2 * K - 1; j++) {
                    int l = lane - (K - 1) + j;
                    l = l < 0 ? 0 : (l > WAVE - 1 ? WAVE - 1 : l);
                    qs[j] = tile[stile_at<IXY, TA>(m, l, ac)];
                }
                weno_pyweno_k<K>(qs, ql[m], qr[m]);
            } else if (LIM == 1) tvd2_cell(qm1, q0, qp1, a.mthlim[m < MAX_WAVES_K ? m : MAX_WAVES_K - 1], ql[m], qr[m]);
            else if (LIM == 2) weno5_pyweno(qm2, qm1, q0, qp1, qp2, ql[m], qr[m]);
            else weno5_legacy(qm2, qm1, q0, qp1, qp2, ql[m], qr[m]);
        }
        double dtdx_c = a.dtd;
        if constexpr (CAPA) dtdx_c = a.dt / (a.dx * tile[stile_at<IXY, TA>(MEQN, lane, ac)]);  // flux1.f90:60
        const double dtdx_l = CAPA ? from_left(dtdx_c) : dtdx_c;

        // both edge states carry the cell's own aux values (flux1.f90:125 passes aux,aux)
        double auxv[NAUX > 0 ? NAUX : 1];
        if constexpr (IXY == 1 && NAUX > 0) {
            const int cc = ca < n_along ? ca : n_along - 1;
            const long g = (long)gb * a.pitch + cc;
#pragma unroll
            for (int m = 0; m < NAUX; m++) auxv[m] = a.aux[aux_idx<RP, 1>(m) * a.plane + g];
        } else {
#pragma unroll
            for (int m = 0; m < NAUX; m++) auxv[m] = tile[stile_at<IXY, TA>(PAUX + m, lane, ac)];
        }
        Cell cl, cr;
        if constexpr (NAUX > 0) {
            cl = RP::template precell<IXY>(ql, a.par, auxv);
            cr = RP::template precell<IXY>(qr, a.par, auxv);
        } else {
            cl = RP::template precell<IXY>(ql, a.par);   // left-edge state of this cell
            cr = RP::template precell<IXY>(qr, a.par);   // right-edge state of this cell
        }
        const Cell crl = struct_from_left(cr);                  // right-edge state of cell i-1
        double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN], amdq2[MEQN], apdq2[MEQN];
        RP::template solve<IXY>(crl, cl, a.par, wave, s, amdq, apdq);       // interface i   (flux1.f90:125)
        if (cfl_ok) {
#pragma unroll
            for (int mw = 0; mw < MWAVES; mw++)
                cflmax = dmax(dmax(cflmax, dtdx_c * s[mw]), -dtdx_l * s[mw]);
        }
        RP::template solve<IXY>(cl, cr, a.par, wave, s, amdq2, apdq2);      // inside cell i (flux1.f90:182)
        double dq1[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++) {  // the shift must run with every lane active
            const double amdq_r = from_right(amdq[m]);
            dq1[m] = -(dtdx_c * (amdq_r + apdq[m] + amdq2[m] + apdq2[m]));
        }
        if (owned) {
#pragma unroll
            for (int m = 0; m < MEQN; m++) tile[stile_at<IXY, TA>(m, lane, ac)] = dq1[m];
        }
    }
    __syncthreads();

    // cooperative store of dq for the owned interior cells
    // dq/6 of the SSP33 / Euler-substage combination (op 1): one shared reciprocal, correctly rounded quotients
    // (rp.hpp Recip; dq is a normal-range quantity or exactly zero) instead of 20 IEEE divisions per thread
    const Recip by_ca(a.rk_op == 1 ? a.rk_ca : 1.0);
#pragma unroll
    for (int k = 0; k < NLD; k++) {
        const int al = l_al + k * STEP_AL, ac = l_ac + k * STEP_AC;
        const int ga = a0 + al, gb = b0 + ac;
        const bool inner_al = (ga >= a.mbc) && (ga < a.mbc + m_along) && al >= SHALO && al < WAVE - SHALO;
        const bool inner_ac = gb >= 0 && gb < n_across && (one_d || ((gb >= a.mbc) && (gb < a.mbc + m_across)));
        if (inner_al && inner_ac) {
            const long g = IXY == 1 ? (long)gb * a.pitch + ga : (long)ga * a.pitch + gb;
            // SRC: deltaq += dq_src(stage) (sharpclaw.py:232-235) with the device twin of the app's dq_Euler_radial,
            // evaluated on the stage's own q; a separate instantiation, the others carry none of this
            double dsrc[MEQN];
            if constexpr (SRC) {
                static_assert(IXY == 2 && MEQN >= 4, "the fused dq source belongs to the last pass of the 2-D Euler stage");
                double d4[4];
                euler_radial_dq(a.qin[g], a.qin[a.plane + g], a.qin[2 * a.plane + g], a.qin[3 * a.plane + g], a.aux[g],
                                a.dt, a.src_p[0], a.src_p[1], d4);
#pragma unroll
                for (int m = 0; m < MEQN; m++) dsrc[m] = m < 4 ? d4[m] : 0.0;
            }
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                const long at = m * a.plane + g;
                const double v = tile[stile_at<IXY, TA>(m, al, ac)];
                double dq = IXY == 1 ? v : a.qout[at] + v;  // dq = (0 + dq1d_x) + dq1d_y
                if constexpr (SRC) dq = dq + dsrc[m];
                // last pass of a stage: the RK combination of sharpclaw.py:168-206 (same expressions as rk_kernel).
                // Branch-free on purpose: with a scalar branch per op the ROCm 7.2 backend left the store base
                // of the op-5 path undefined in the 1-D instantiation (memory fault at address 0).
#ifdef PCL_SHARP_BRANCHY   /* diagnostic build only: the form that faulted in round 1 */
                switch (a.rk_op) {
                case 0: a.qout[at] = dq; break;
                case 1: a.rk_d[at] = a.rk_a[at] + dq / a.rk_ca; break;
                case 2: a.rk_d[at] = a.rk_ca * a.rk_a[at] + a.rk_cb * (a.rk_b[at] + dq); break;
                case 5: a.rk_d[at] = a.rk_a[at] + a.rk_cb * a.rk_b[at] + a.rk_cc * dq; break;
                }
#else
                double r = dq;
                if (a.rk_op != 0) {
                    const double av = a.rk_a[at], bv = a.rk_b[at];
                    double quo = by_ca.div(dq);
#if !PCL_FAST
                    // an increment deep in the underflow range (tails of a source term, products of tiny momenta): the
                    // shortcut's residual is no longer exact there, take the IEEE quotient (rare, divergent branch)
                    // (the empty volatile asm keeps it a branch: if-converted, the division would run for every cell
                    // and cost the pass 4 %)
                    if (a.rk_op == 1 && __builtin_fabs(dq) < 0x1p-900 && dq != 0.0) {
                        asm volatile("");
                        quo = dq / a.rk_ca;
                    }
#endif
                    const double r1 = av + quo;
                    const double r2 = a.rk_ca * av + a.rk_cb * (bv + dq);
                    const double r5 = av + a.rk_cb * bv + a.rk_cc * dq;
                    r = a.rk_op == 1 ? r1 : (a.rk_op == 2 ? r2 : r5);
                }
                double *dst = a.rk_op != 0 ? a.rk_d : a.qout;
                dst[at] = r;
#endif
            }
        }
    }
    cfl_publish(a.cfl, cflmax);
}

// ---- char_decomp = 1 in 1-D: wave-based reconstruction (1d/sharpclaw/flux1.f90:80-107) -------------------------------
// flux1 first solves the Riemann problems between the cell averages (rp1 on q1d, q1d) and hands the waves to tvd2_wave
// (lim_type 1, reconstruct.f90:728-806) or weno5_wave (lim_type 2, :393-478), which build the two states of every
// INTERFACE from them: q^-(i-1/2) = qr(i-1) and q^+(i-1/2) = ql(i).  One lane = one cell = the interface on its left:
// lane l solves (l-1, l), fetches the waves of the interfaces l-2, l-1, l+1, l+2 by DPP shifts, builds qm = qr(l-1) and
// qp = ql(l), solves the interface problem (qm, qp) and the problem inside the cell (qp, the right lane's qm).  Halo 3
// like WENO5: lanes 3..60 are stored.  Solvers without aux arrays, no capacity function.  Operation order of
// oracle/sharpclaw_oracle.c: weno5_wave / tvd2_wave (== the reference's Fortran bit for bit, tests/golden/ref_recon_wave.npz).
template <int MEQN>
__device__ __forceinline__ double dot_m(const double (&a)[MEQN], const double (&b)[MEQN]) {
    double d = a[0] * b[0];
#pragma unroll
    for (int m = 1; m < MEQN; m++) d = d + a[m] * b[m];
    return d;
}
// one wave family of weno5_wave: the waves at interfaces i-2 .. i+2 -> the coefficients u(1), u(2) and 1/|w|^2 (or 0)
template <int MEQN>
__device__ __forceinline__ void weno5_wave_family(const double (&wm2)[MEQN], const double (&wm1)[MEQN], const double (&w0)[MEQN],
                                                  const double (&wp1)[MEQN], const double (&wp2)[MEQN], double &u1, double &u2,
                                                  double &winv) {
    const double epweno = PCL_F32(1.e-36), tol = PCL_F32(1.e-14);
    double u[2], wnorm2 = 0.0;
#pragma unroll
    for (int m1 = 0; m1 < 2; m1++) {
        const double im = m1 == 0 ? 1.0 : -1.0;
        wnorm2 = dot_m<MEQN>(w0, w0);
        const double theta1 = m1 == 0 ? dot_m<MEQN>(wm2, w0) : dot_m<MEQN>(wp2, w0);     // i + intwo
        const double theta2 = m1 == 0 ? dot_m<MEQN>(wm1, w0) : dot_m<MEQN>(wp1, w0);     // i + inone
        const double theta3 = m1 == 0 ? dot_m<MEQN>(wp1, w0) : dot_m<MEQN>(wm1, w0);     // i + ione
        const double t1 = im * (theta1 - theta2), t2 = im * (theta2 - wnorm2), t3 = im * (wnorm2 - theta3);
        double x, tt1, tt2, tt3;
        x = theta1 - 3. * theta2; tt1 = 13. * (t1 * t1) + 3. * (x * x);
        x = theta2 + wnorm2;      tt2 = 13. * (t2 * t2) + 3. * (x * x);
        x = 3. * wnorm2 - theta3; tt3 = 13. * (t3 * t3) + 3. * (x * x);
        x = epweno + tt1; tt1 = x * x;
        x = epweno + tt2; tt2 = x * x;
        x = epweno + tt3; tt3 = x * x;
        double s1 = tt2 * tt3;
        const double s2 = 6. * tt1 * tt3;
        double s3 = 3. * tt1 * tt2;
        const double t0 = fdiv_ieee(1.0, s1 + s2 + s3);
        s1 = s1 * t0;
        s3 = s3 * t0;
        if (wnorm2 > tol) {
            u[m1] = fdiv_ieee(s1 * (t2 - t1) + (0.5 * s3 - 0.25) * (t3 - t2), 3.0);
            wnorm2 = fdiv_ieee(1.0, wnorm2);
        } else {
            u[m1] = 0.0;
            wnorm2 = 0.0;
        }
    }
    u1 = u[0]; u2 = u[1]; winv = wnorm2;
}
// the limiter of tvd2_wave (reconstruct.f90:771-793): philim's 1..4 and Cada & Torrilhon's (5)
__device__ __forceinline__ double tvd2_wave_limiter(double r, int meth) {
    switch (meth) {
    case 1: return dmax(0.0, dmin(1.0, r));
    case 2: return dmax(dmax(0.0, dmin(1.0, 2.0 * r)), dmin(2.0, r));
    case 3: return fdiv_ieee(r + fabs(r), 1.0 + fabs(r));
    case 4: { const double c = (1.0 + r) / 2.0; return dmax(0.0, dmin(dmin(c, 2.0), 2.0 * r)); }
    case 5: {
        const double alpha = fdiv_ieee(1.0, 3.0);
        const double pp = fdiv_ieee(2.0 + r, 3.0);
        const double amax = dmax(dmax(-alpha * r, 0.0), dmin(dmin(2.0 * r, pp), 2.0));
        return dmax(0.0, dmin(pp, amax));
    }
    }
    return 0.0;
}

template <class RP, int LIM>
__global__ __launch_bounds__(256) void sharp1w_kernel(SweepArgs a, int nstrips) {
    constexpr int MEQN = RP::MEQN, MWAVES = RP::MWAVES, SH = 3, SS = sstrip(SH);
    static_assert(RP::NAUX == 0, "wave-based reconstruction: solvers without aux arrays");
    static_assert(LIM == 1 || LIM == 2, "tvd2_wave / weno5_wave");
    using Cell = typename RP::Cell;
    const int lane = threadIdx.x & (WAVE - 1);
    const int strip = blockIdx.x * (256 / WAVE) + threadIdx.x / WAVE;
    if (strip >= nstrips) return;                         // wave-uniform (no barrier in this kernel)
    const int a0 = a.mbc - SH + strip * SS;
    const int ca = a0 + lane;
    const int cc = ca < a.I ? ca : a.I - 1;
    const bool owned = (ca >= a.mbc) && (ca < a.mbc + a.mx) && lane >= SH && lane < WAVE - SH;
    const bool cfl_ok = (ca >= a.mbc) && (ca <= a.mbc + a.mx) && lane >= SH && lane <= WAVE - SH;
    double q[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) q[m] = a.qin[m * a.plane + cc];
    // rp1(q1d, q1d): the waves between the cell averages, interface `lane` = (lane-1, lane)
    const Cell c0 = RP::template precell<1>(q, a.par);
    const Cell cm = struct_from_left(c0);
    double w0[MWAVES][MEQN], s0[MWAVES], t_am[MEQN], t_ap[MEQN];
    RP::template solve<1>(cm, c0, a.par, w0, s0, t_am, t_ap);
    double qm[MEQN], qp[MEQN];
    const double qm1c = 0.0;
    (void)qm1c;
    double ql1[MEQN];            // the left cell's average
#pragma unroll
    for (int m = 0; m < MEQN; m++) ql1[m] = from_left(q[m]);
    if constexpr (LIM == 2) {
        // stencil-independent part (reconstruct.f90:415-418): (-q(i-2) + 7 (q(i-1) + q(i)) - q(i+1)) / 12
#pragma unroll
        for (int m = 0; m < MEQN; m++) {
            const double qm2 = from_left(ql1[m]), qp1 = from_right(q[m]);
            qm[m] = fdiv_ieee(-qm2 + 7. * (ql1[m] + q[m]) - qp1, 12.0);
            qp[m] = qm[m];
        }
    } else {
#pragma unroll
        for (int m = 0; m < MEQN; m++) { qm[m] = ql1[m]; qp[m] = q[m]; }
    }
#pragma unroll
    for (int mw = 0; mw < MWAVES; mw++) {
        double wm1[MEQN], wp1[MEQN];
#pragma unroll
        for (int m = 0; m < MEQN; m++) { wm1[m] = from_left(w0[mw][m]); wp1[m] = from_right(w0[mw][m]); }
        if constexpr (LIM == 2) {
            double wm2[MEQN], wp2[MEQN], u1, u2, winv;
#pragma unroll
            for (int m = 0; m < MEQN; m++) { wm2[m] = from_left(wm1[m]); wp2[m] = from_right(wp1[m]); }
            weno5_wave_family<MEQN>(wm2, wm1, w0[mw], wp1, wp2, u1, u2, winv);
#pragma unroll
            for (int m = 0; m < MEQN; m++) {
                qm[m] = qm[m] + u1 * w0[mw][m] * winv;
                qp[m] = qp[m] + u2 * w0[mw][m] * winv;
            }
        } else {
            // tvd2_wave: dotl = w(i-1).w(i), dotr = w(i).w(i+1), sums started from 0.d0 (same values)
            const double wnorm2 = dot_m<MEQN>(w0[mw], w0[mw]);
            const double dotr = dot_m<MEQN>(w0[mw], wp1);
            const double dotl = dot_m<MEQN>(wm1, w0[mw]);
            if (wnorm2 != 0.0) {
                const double r = fdiv_ieee(s0[mw] > 0.0 ? dotl : dotr, wnorm2);
                const double uu = 0.5 * tvd2_wave_limiter(r, a.mthlim[mw]);
#pragma unroll
                for (int m = 0; m < MEQN; m++) {
                    qm[m] = qm[m] + w0[mw][m] * uu;
                    qp[m] = qp[m] - w0[mw][m] * uu;
                }
            }
        }
    }
    // interface problem (qr(i-1), ql(i)) and the problem inside the cell (ql(i), qr(i)): flux1.f90:125-187
    const Cell em = RP::template precell<1>(qm, a.par), ep = RP::template precell<1>(qp, a.par);
    double wave[MWAVES][MEQN], s[MWAVES], amdq[MEQN], apdq[MEQN], amdq2[MEQN], apdq2[MEQN];
    RP::template solve<1>(em, ep, a.par, wave, s, amdq, apdq);
    double cflmax = 0.0;
    if (cfl_ok) {
#pragma unroll
        for (int mw = 0; mw < MWAVES; mw++) cflmax = dmax(dmax(cflmax, a.dtd * s[mw]), -a.dtd * s[mw]);
    }
    double qmr[MEQN];
#pragma unroll
    for (int m = 0; m < MEQN; m++) qmr[m] = from_right(qm[m]);          // qr(i) = the right lane's q^-
    const Cell er = RP::template precell<1>(qmr, a.par);
    RP::template solve<1>(ep, er, a.par, wave, s, amdq2, apdq2);
    const Recip by_ca(a.rk_op == 1 ? a.rk_ca : 1.0);
#pragma unroll
    for (int m = 0; m < MEQN; m++) {      // the shift must run with every lane active
        const double amdq_r = from_right(amdq[m]);
        const double dq = -(a.dtd * (amdq_r + apdq[m] + amdq2[m] + apdq2[m]));
        if (owned) {
            const long at = m * a.plane + cc;
            double r = dq;
            if (a.rk_op != 0) {       // the RK combination fused into the (only) pass: sharp_kernel's store phase
                const double av = a.rk_a[at], bv = a.rk_b[at];
                double quo = by_ca.div(dq);
#if !PCL_FAST
                if (a.rk_op == 1 && __builtin_fabs(dq) < 0x1p-900 && dq != 0.0) { asm volatile(""); quo = dq / a.rk_ca; }
#endif
                const double r1 = av + quo;
                const double r2 = a.rk_ca * av + a.rk_cb * (bv + dq);
                const double r5 = av + a.rk_cb * bv + a.rk_cc * dq;
                r = a.rk_op == 1 ? r1 : (a.rk_op == 2 ? r2 : r5);
            }
            double *dst = a.rk_op != 0 ? a.rk_d : a.qout;
            dst[at] = r;
        }
    }
    cfl_publish(a.cfl, cflmax);
}

// ---- Runge-Kutta register arithmetic (sharpclaw.py:168-206), elementwise over whole arrays ----------
struct RkOp {
    double *d;
    const double *a, *b, *c;
    double ca, cb, cc;
    long n;
    int op;
};
__global__ __launch_bounds__(256) void rk_kernel(RkOp o) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < o.n; i += (long)gridDim.x * blockDim.x) {
        double r;
        switch (o.op) {
        case 1: r = o.a[i] + o.b[i] / o.ca; break;                          // A + B/c
        case 2: r = o.ca * o.a[i] + o.cb * (o.b[i] + o.c[i]); break;        // a*A + b*(B + C)
        case 3: r = o.a[i] / o.ca + o.cb * o.b[i]; break;                   // A/c + b*B
        case 4: r = o.ca * o.a[i] - o.cb * o.b[i]; break;                   // a*A - b*B
        case 5: r = o.a[i] + o.cb * o.b[i] + o.cc * o.c[i]; break;          // A + b*B + c*C
        case 6: {  // SSP104 mid-step pair in one pass (sharpclaw.py:186-187): C = A/ca + cb*B ; D = cc*C - 5*B
            const double s2 = o.a[i] / o.ca + o.cb * o.b[i];
            const_cast<double *>(o.c)[i] = s2;
            r = o.cc * s2 - 5. * o.b[i];
            break;
        }
        default: r = o.a[i];
        }
        o.d[i] = r;
    }
}

}  // namespace PCL_NS
}  // namespace pcl
