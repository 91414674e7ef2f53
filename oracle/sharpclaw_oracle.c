/*
 * oracle/sharpclaw_oracle.c -- TEST INFRASTRUCTURE ONLY (parity oracle, see classic_oracle.c).
 *
 * C restatement of the reference's SharpClaw semi-discrete right-hand side:
 *   flux2 (2-D slice driver)  src/fortran/2d/sharpclaw/flux2.f90:32-94
 *   flux1                     src/fortran/2d/sharpclaw/flux1.f90:59-188 (1-D twin 1d/sharpclaw/flux1.f90)
 *   weno5 (PyWENO-generated)  src/fortran/1d/sharpclaw/weno.f90:36-100   lim_type=2, char_decomp=0
 *   weno5 (legacy)            src/fortran/1d/sharpclaw/reconstruct.f90:120-185   lim_type=3
 *   tvd2                      src/fortran/1d/sharpclaw/reconstruct.f90:568-625   lim_type=1, char_decomp=0
 * Only char_decomp=0, tfluct_solver=.false. (the configurations the reference's tests use).
 *
 * The PyWENO source writes its constants WITHOUT a d0 exponent (+3.33333333333333, +0.1, 1.0e-36):
 * they are REAL*4 literals promoted to double; the legacy routine's epweno = 1.e-36 likewise.
 * F32() below reproduces that (float32-rounded constants => diff 0.0 vs flang).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

int orc_rpn2_ptr(int rp, const double *par, int ixy, int meqn, int mwaves, int mbc, int mx,
                 const double *ql, const double *qr, double *wave, double *s, double *amdq, double *apdq);
int orc_rp1_ptr(int rp, const double *par, int meqn, int mwaves, int mbc, int mx, const double *ql,
                const double *qr, double *wave, double *s, double *amdq, double *apdq);

#define F32(x) ((double)(float)(x))

static inline double dmax(double a, double b) { return a > b ? a : b; }

/* q, ql, qr: (meqn, n) Fortran order, 1-based second index i = 1..n in the Fortran; here 0-based k=i-1 */
#define Q(m, i) q[(m) + meqn * ((i)-1)]
#define QL(m, i) ql[(m) + meqn * ((i)-1)]
#define QR(m, i) qr[(m) + meqn * ((i)-1)]

/* weno.f90:36-100; loop i = mbc-1 .. maxnx+mbc+1 restricted to indices whose stencil is in range */
static void weno5_pyweno(const double *q, double *ql, double *qr, int meqn, int n, int ilo, int ihi)
{
    const double c333 = F32(+3.33333333333333), c1033 = F32(-10.3333333333333), c366 = F32(+3.66666666666667);
    const double c833 = F32(+8.33333333333333), c633 = F32(-6.33333333333333), c133 = F32(+1.33333333333333);
    const double c433m = F32(-4.33333333333333), c166 = F32(+1.66666666666667), c433 = F32(+4.33333333333333);
    const double eps = F32(1.0e-36), w01 = F32(+0.1), w06 = F32(+0.6), w03 = F32(+0.3);
    const double f183 = F32(+1.83333333333333), f116 = F32(-1.16666666666667), f033 = F32(+0.333333333333333);
    const double f083 = F32(+0.833333333333333), f016 = F32(-0.166666666666667);
    (void)n;
    for (int i = ilo; i <= ihi; i++)
        for (int m = 0; m < meqn; m++) {
            double qm2 = Q(m, i - 2), qm1 = Q(m, i - 1), q0 = Q(m, i), qp1 = Q(m, i + 1), qp2 = Q(m, i + 2);
            double sigma0 = ((c333 * q0) * q0) + ((c1033 * q0) * qp1) + ((c366 * q0) * qp2) +
                            ((c833 * qp1) * qp1) + ((c633 * qp1) * qp2) + ((c133 * qp2) * qp2);
            double sigma1 = ((c133 * qm1) * qm1) + ((c433m * qm1) * q0) + ((c166 * qm1) * qp1) +
                            ((c433 * q0) * q0) + ((c433m * q0) * qp1) + ((c133 * qp1) * qp1);
            double sigma2 = ((c133 * qm2) * qm2) + ((c633 * qm2) * qm1) + ((c366 * qm2) * q0) +
                            ((c833 * qm1) * qm1) + ((c1033 * qm1) * q0) + ((c333 * q0) * q0);
            double acc = 0.0, t;
            t = sigma0 + eps; double omega0 = w01 / (t * t); acc = acc + omega0;
            t = sigma1 + eps; double omega1 = w06 / (t * t); acc = acc + omega1;
            t = sigma2 + eps; double omega2 = w03 / (t * t); acc = acc + omega2;
            omega0 = omega0 / acc; omega1 = omega1 / acc; omega2 = omega2 / acc;
            acc = 0.0;
            t = sigma0 + eps; double omega3 = w03 / (t * t); acc = acc + omega3;
            t = sigma1 + eps; double omega4 = w06 / (t * t); acc = acc + omega4;
            t = sigma2 + eps; double omega5 = w01 / (t * t); acc = acc + omega5;
            omega3 = omega3 / acc; omega4 = omega4 / acc; omega5 = omega5 / acc;
            double fr0 = f183 * q0 + f116 * qp1 + f033 * qp2;
            double fr1 = f033 * qm1 + f083 * q0 + f016 * qp1;
            double fr2 = f016 * qm2 + f083 * qm1 + f033 * q0;
            double fr3 = f033 * q0 + f083 * qp1 + f016 * qp2;
            double fr4 = f016 * qm1 + f083 * q0 + f033 * qp1;
            double fr5 = f033 * qm2 + f116 * qm1 + f183 * q0;
            QL(m, i) = omega0 * fr0 + omega1 * fr1 + omega2 * fr2;
            QR(m, i) = omega3 * fr3 + omega4 * fr4 + omega5 * fr5;
        }
}

/* weno.f90: weno7 ... weno17 (and weno5 again, k = 3): the generated subroutines all have one shape -- for the k
 * stencils r = 0..k-1 (cells i-r .. i-r+k-1) the smoothness indicator as a sum over a <= b of ((C*q_a)*q_b) in
 * lexicographic order, the two sets of nonlinear weights (w / (sigma + 1e-36)**2, normalised by their running sum),
 * the 2k candidate values and their weighted sums, every sum left-associated in the printed order.  The coefficients
 * are derived from first principles by tools/gen_weno.py (and compared there with the literals of the reference's
 * file); like the Fortran's REAL*4 literals they are float32-rounded.  order = 2k-1, mbc = k. */
#include "weno_tables.h"
static int orc_weno_order = 5;
void orc_sharp_set_weno_order(int order) { orc_weno_order = order; }
static void weno_pyweno_k(int k, const double *q, double *ql, double *qr, int meqn, int n)
{
    const int t = k - 3;
    const double eps = F32(1.0e-36);
    for (int i = k; i <= n - (k - 1); i++)          /* every index whose (2k-1)-point stencil is in range */
        for (int m = 0; m < meqn; m++) {
            double sigma[9], oml[9], omr[9];
            for (int r = 0; r < k; r++) {
                double sg = 0.0;
                int first = 1;
                for (int a = 0; a < k; a++)
                    for (int b = a; b < k; b++) {
                        const double term = (WENO_SIG[t][r][a][b] * Q(m, i - r + a)) * Q(m, i - r + b);
                        sg = first ? term : sg + term;
                        first = 0;
                    }
                sigma[r] = sg;
            }
            double acc = 0.0;
            for (int r = 0; r < k; r++) {
                const double tt = sigma[r] + eps;
                oml[r] = WENO_WL[t][r] / (tt * tt);
                acc = acc + oml[r];
            }
            for (int r = 0; r < k; r++) oml[r] = oml[r] / acc;
            acc = 0.0;
            for (int r = 0; r < k; r++) {
                const double tt = sigma[r] + eps;
                omr[r] = WENO_WR[t][r] / (tt * tt);
                acc = acc + omr[r];
            }
            for (int r = 0; r < k; r++) omr[r] = omr[r] / acc;
            double fs0 = 0.0, fs1 = 0.0;
            for (int r = 0; r < k; r++) {
                double fl = 0.0, fr = 0.0;
                for (int j = 0; j < k; j++) {
                    const double tl = WENO_CL[t][r][j] * Q(m, i - r + j), tr = WENO_CR[t][r][j] * Q(m, i - r + j);
                    fl = j == 0 ? tl : fl + tl;
                    fr = j == 0 ? tr : fr + tr;
                }
                fs0 = r == 0 ? oml[r] * fl : fs0 + oml[r] * fl;
                fs1 = r == 0 ? omr[r] * fr : fs1 + omr[r] * fr;
            }
            QL(m, i) = fs0;
            QR(m, i) = fs1;
        }
}

/* reconstruct.f90:120-185.  uu(1,i) -> qr(i-1), uu(2,i) -> ql(i); i = mbc .. mx2-mbc+1 */
static void weno5_legacy(const double *q, double *ql, double *qr, int meqn, int mx2, int mbc)
{
    const double epweno = F32(1.e-36);
    double *dq1m = calloc((size_t)mx2 + 4, sizeof(double));
#define DQ(i) dq1m[(i)]
    for (int m = 0; m < meqn; m++) {
        for (int i = 2; i <= mx2; i++) DQ(i) = Q(m, i) - Q(m, i - 1);
        for (int m1 = 1; m1 <= 2; m1++) {
            int im = (m1 == 1) ? 1 : -1;
            int ione = im, inone = -im, intwo = -2 * im;
            for (int i = mbc; i <= mx2 - mbc + 1; i++) {
                double t1 = im * (DQ(i + intwo) - DQ(i + inone));
                double t2 = im * (DQ(i + inone) - DQ(i));
                double t3 = im * (DQ(i) - DQ(i + ione));
                double a1 = DQ(i + intwo) - 3. * DQ(i + inone);
                double a2 = DQ(i + inone) + DQ(i);
                double a3 = 3. * DQ(i) - DQ(i + ione);
                double tt1 = 13. * (t1 * t1) + 3. * (a1 * a1);
                double tt2 = 13. * (t2 * t2) + 3. * (a2 * a2);
                double tt3 = 13. * (t3 * t3) + 3. * (a3 * a3);
                tt1 = (epweno + tt1) * (epweno + tt1);
                tt2 = (epweno + tt2) * (epweno + tt2);
                tt3 = (epweno + tt3) * (epweno + tt3);
                double s1 = tt2 * tt3;
                double s2 = 6. * tt1 * tt3;
                double s3 = 3. * tt1 * tt2;
                double t0 = 1. / (s1 + s2 + s3);
                s1 = s1 * t0;
                s3 = s3 * t0;
                double uu = (s1 * (t2 - t1) + (0.5 * s3 - 0.25) * (t3 - t2)) / 3. +
                            (-Q(m, i - 2) + 7. * (Q(m, i - 1) + Q(m, i)) - Q(m, i + 1)) / 12.;
                if (m1 == 1) QR(m, i - 1) = uu;
                else QL(m, i) = uu;
            }
        }
    }
#undef DQ
    free(dq1m);
}

/* reconstruct.f90:568-625 (tvd2): second-order TVD reconstruction, component-wise; mthlim is indexed by COMPONENT
 * (select case(mthlim(m)), :594).  Two things a maintainer should know about the reference routine:
 *  * `dqm = dqp` (:590) runs before dqp has ever been assigned: at the first cell of every component the Fortran
 *    uses an UNINITIALISED value (for m > 1 the last dqp of the previous component; for m = 1 whatever the stack
 *    holds).  That only reaches ql/qr of ghost cell 0 and through it the first interior cell of each slice.  Here
 *    the first cell gets the intended dqm = q(i) - q(i-1); the reference-generated golden is therefore compared
 *    away from the first interior row / column (tests/test_ref_goldens.py).
 *  * on locally constant data r = 0/0.  gfortran's MAX/MIN (the reference's compiler) return the non-NaN operand,
 *    so qlimitr is finite and ql = qr = q; flang's propagate the NaN (oracle/_ref built here returns NaN on a
 *    constant patch).  fmax/fmin below = the gfortran behaviour, which is also what v_max/min_f64 do on the GPU.
 * The hard-wired mbc = 2 of the routine means the loop runs over slice indices 3 .. n-2. */
static int orc_tvd_mthlim[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1};
void orc_sharp_set_mthlim(const int *mthlim, int n)
{
    for (int k = 0; k < 16; k++) orc_tvd_mthlim[k] = (k < n) ? mthlim[k] : 1;
}
static void tvd2(const double *q, double *ql, double *qr, int meqn, int n)
{
    for (int m = 0; m < meqn; m++) {
        double dqp = Q(m, 3) - Q(m, 2);
        for (int i = 3; i <= n - 2; i++) {
            const double dqm = dqp;
            dqp = Q(m, i + 1) - Q(m, i);
            const double r = dqp / dqm;
            double qlimitr = 0.0;
            switch (orc_tvd_mthlim[m]) {
            case 1: qlimitr = fmax(0.0, fmin(1.0, r)); break;
            case 2: qlimitr = fmax(fmax(0.0, fmin(1.0, 2.0 * r)), fmin(2.0, r)); break;
            case 3: qlimitr = (r + fabs(r)) / (1.0 + fabs(r)); break;
            case 4: { const double c = (1.0 + r) / 2.0; qlimitr = fmax(0.0, fmin(fmin(c, 2.0), 2.0 * r)); break; }
            case 5: {
                const double beta = 2.0, xgamma = 2.0, alpha = 1.0 / 3.0;
                const double pp = (2.0 + r) / 3.0;
                const double amax = fmax(fmax(-alpha * r, 0.0), fmin(fmin(beta * r, pp), xgamma));
                qlimitr = fmax(0.0, fmin(pp, amax));
                break;
            }
            }
            QR(m, i) = Q(m, i) + 0.5 * qlimitr * dqm;
            QL(m, i) = Q(m, i) - 0.5 * qlimitr * dqm;
        }
    }
}

/* ---- char_decomp = 1: wave-based reconstructions (1d/sharpclaw/reconstruct.f90) ------------------------------------
 * q (meqn, n), wave (meqn, mwaves, n), s (mwaves, n), Fortran order, 1-based second / third index.  The Fortran loops
 * run over every interface i = 2..n and read wave(:,:,i-2..i+2) and q(:,i-2..i+1): the first and last ones reach
 * outside the arrays (reconstruct.f90:411-470; harmless for the interior, mbc = 3).  Here an index outside 1..n reads
 * 0 -- the golden generator pads its arrays with zeros accordingly -- so results agree at EVERY index.
 * REAL*4 literals (7., 12., 13., 3., 6., 0.5, 0.25, 1./..., 1.e-14, epweno = 1.e-36) are promoted from float. */
static int orc_char_decomp = 0, orc_sharp_fwave = 0;
void orc_sharp_set_char_decomp(int cd, int fwave) { orc_char_decomp = cd; orc_sharp_fwave = fwave; }
#define WV(m, mw, i) (((i) < 1 || (i) > n) ? 0.0 : wave[(m) + meqn * ((mw) + (size_t)mwaves * ((i)-1))])
#define QZ(m, i) (((i) < 1 || (i) > n) ? 0.0 : q[(m) + meqn * ((i)-1)])
#define SV(mw, i) s[(mw) + mwaves * ((i)-1)]

/* reconstruct.f90:393-478 (fw = 0) and :481-565 (fw = 1; the caller has divided the f-waves by s) */
static void weno5_wave(const double *q, double *ql, double *qr, const double *wave, int meqn, int mwaves, int n, int fw)
{
    const double epweno = (double)1.e-36f, tol = (double)1.e-14f;
    for (int i = 2; i <= n; i++) {
        for (int m = 0; m < meqn; m++) {
            if (fw) { QR(m, i - 1) = QZ(m, i - 1); QL(m, i) = QZ(m, i); }
            else {
                QR(m, i - 1) = (-QZ(m, i - 2) + 7. * (QZ(m, i - 1) + QZ(m, i)) - QZ(m, i + 1)) / 12.;
                QL(m, i) = QR(m, i - 1);
            }
        }
        for (int mw = 0; mw < mwaves; mw++) {
            double u[2], wnorm2 = 0.0;
            for (int m1 = 1; m1 <= 2; m1++) {
                const int im = (m1 == 1) ? 1 : -1;
                const int ione = im, inone = -im, intwo = -2 * im;
                double theta1, theta2, theta3;
                wnorm2 = WV(0, mw, i) * WV(0, mw, i);
                theta1 = WV(0, mw, i + intwo) * WV(0, mw, i);
                theta2 = WV(0, mw, i + inone) * WV(0, mw, i);
                theta3 = WV(0, mw, i + ione) * WV(0, mw, i);
                for (int m = 1; m < meqn; m++) {
                    wnorm2 = wnorm2 + WV(m, mw, i) * WV(m, mw, i);
                    theta1 = theta1 + WV(m, mw, i + intwo) * WV(m, mw, i);
                    theta2 = theta2 + WV(m, mw, i + inone) * WV(m, mw, i);
                    theta3 = theta3 + WV(m, mw, i + ione) * WV(m, mw, i);
                }
                const double t1 = im * (theta1 - theta2);
                const double t2 = im * (theta2 - wnorm2);
                const double t3 = im * (wnorm2 - theta3);
                double a, tt1, tt2, tt3;
                a = theta1 - 3. * theta2; tt1 = 13. * (t1 * t1) + 3. * (a * a);
                a = theta2 + wnorm2;      tt2 = 13. * (t2 * t2) + 3. * (a * a);
                a = 3. * wnorm2 - theta3; tt3 = 13. * (t3 * t3) + 3. * (a * a);
                a = epweno + tt1; tt1 = a * a;
                a = epweno + tt2; tt2 = a * a;
                a = epweno + tt3; tt3 = a * a;
                double s1 = tt2 * tt3;
                const double s2 = 6. * tt1 * tt3;
                double s3 = 3. * tt1 * tt2;
                const double t0 = 1. / (s1 + s2 + s3);
                s1 = s1 * t0;
                s3 = s3 * t0;
                if (wnorm2 > tol) {
                    if (fw)
                        u[m1 - 1] = ((s1 * (t2 - t1) + (0.5 * s3 - 0.25) * (t3 - t2)) / 3. +
                                     im * (theta2 + 6.0 * wnorm2 - theta3) / 12.0);
                    else
                        u[m1 - 1] = (s1 * (t2 - t1) + (0.5 * s3 - 0.25) * (t3 - t2)) / 3.;
                    wnorm2 = 1.0 / wnorm2;
                } else {
                    u[m1 - 1] = 0.0;
                    wnorm2 = 0.0;
                }
            }
            for (int m = 0; m < meqn; m++) {
                QR(m, i - 1) = QR(m, i - 1) + u[0] * WV(m, mw, i) * wnorm2;
                QL(m, i) = QL(m, i) + u[1] * WV(m, mw, i) * wnorm2;
            }
        }
    }
}

/* reconstruct.f90:728-806; its local mbc is 2: interfaces i = 2 .. n-2 */
static void tvd2_wave(const double *q, double *ql, double *qr, const double *wave, const double *s, const int *mthlim,
                      int meqn, int mwaves, int n)
{
    for (int i = 2; i <= n; i++)
        for (int m = 0; m < meqn; m++) { QR(m, i - 1) = Q(m, i - 1); QL(m, i) = Q(m, i); }
    for (int mw = 0; mw < mwaves; mw++) {
        double dotr = 0.0;
        for (int i = 2; i <= n - 2; i++) {
            double wnorm2 = 0.0;
            const double dotl = dotr;
            dotr = 0.0;
            for (int m = 0; m < meqn; m++) {
                wnorm2 = wnorm2 + WV(m, mw, i) * WV(m, mw, i);
                dotr = dotr + WV(m, mw, i) * WV(m, mw, i + 1);
            }
            if (wnorm2 == 0.0) continue;
            const double r = (SV(mw, i) > 0.0) ? dotl / wnorm2 : dotr / wnorm2;
            double wlimitr = 0.0;
            switch (mthlim[mw]) {
            case 1: wlimitr = fmax(0.0, fmin(1.0, r)); break;
            case 2: wlimitr = fmax(fmax(0.0, fmin(1.0, 2.0 * r)), fmin(2.0, r)); break;
            case 3: wlimitr = (r + fabs(r)) / (1.0 + fabs(r)); break;
            case 4: { const double c = (1.0 + r) / 2.0; wlimitr = fmax(0.0, fmin(fmin(c, 2.0), 2.0 * r)); break; }
            case 5: {
                const double beta = 2.0, xgamma = 2.0, alpha = 1.0 / 3.0;
                const double pp = (2.0 + r) / 3.0;
                const double amax = fmax(fmax(-alpha * r, 0.0), fmin(fmin(beta * r, pp), xgamma));
                wlimitr = fmax(0.0, fmin(pp, amax));
                break;
            }
            }
            const double uu = 0.5 * wlimitr;
            for (int m = 0; m < meqn; m++) {
                QR(m, i - 1) = QR(m, i - 1) + WV(m, mw, i) * uu;
                QL(m, i) = QL(m, i) - WV(m, mw, i) * uu;
            }
        }
    }
}

/* direct entry (goldens of the reconstructions themselves): kind 1 tvd2_wave, 2 weno5_wave, 3 weno5_fwave (wave /= s first) */
int orc_recon_wave(int kind, int meqn, int mwaves, int n, const double *q, double *wave, const double *s, const int *mthlim,
                   double *ql, double *qr)
{
    if (kind == 1) tvd2_wave(q, ql, qr, wave, s, mthlim, meqn, mwaves, n);
    else if (kind == 2) weno5_wave(q, ql, qr, wave, meqn, mwaves, n, 0);
    else if (kind == 3) {
        for (int i = 1; i <= n; i++)
            for (int mw = 0; mw < mwaves; mw++)
                for (int m = 0; m < meqn; m++)
                    wave[m + meqn * (mw + (size_t)mwaves * (i - 1))] = wave[m + meqn * (mw + (size_t)mwaves * (i - 1))] / SV(mw, i);
        weno5_wave(q, ql, qr, wave, meqn, mwaves, n, 1);
    } else return -1;
    return 0;
}

/* flux1.f90:59-188 on one slice.  q1d (meqn, 1-mbc:mx+mbc); dq1d same extent, returned. */
static int flux1(int ndim, int rp, const double *par, int lim_type, int ixy, int meqn, int mwaves, int mbc,
                 int mx, const double *q1d, double *dq1d, const double *dtdx, double *cfl_out, double *work)
{
    const int n = mx + 2 * mbc;
    double *ql = work, *qr = ql + (size_t)meqn * n, *wave = qr + (size_t)meqn * n;
    double *s = wave + (size_t)meqn * mwaves * n, *amdq = s + (size_t)mwaves * n;
    double *apdq = amdq + (size_t)meqn * n, *amdq2 = apdq + (size_t)meqn * n, *apdq2 = amdq2 + (size_t)meqn * n;
    memset(work, 0, sizeof(double) * ((size_t)meqn * n * 6 + (size_t)meqn * mwaves * n + (size_t)mwaves * n));
    const double *q = q1d;
    /* the Fortran indexes the slice 1..maxnx+2mbc inside weno: i_weno = i_cell + mbc */
    if (orc_char_decomp == 1) {
        /* 1d/sharpclaw/flux1.f90:80-107: rp1 on (q1d, q1d), then the waves as slopes */
        if (ndim != 1 || (lim_type != 1 && lim_type != 2)) return -4;
        int rcw = orc_rp1_ptr(rp, par, meqn, mwaves, mbc, mx, q, q, wave, s, amdq, apdq);
        if (rcw) return rcw;
        if (lim_type == 1) tvd2_wave(q, ql, qr, wave, s, orc_tvd_mthlim, meqn, mwaves, n);
        else if (orc_sharp_fwave) {
            for (int i = 1; i <= n; i++)
                for (int mw = 0; mw < mwaves; mw++)
                    for (int m = 0; m < meqn; m++)
                        wave[m + meqn * (mw + (size_t)mwaves * (i - 1))] /= s[mw + mwaves * (i - 1)];
            weno5_wave(q, ql, qr, wave, meqn, mwaves, n, 1);
        } else weno5_wave(q, ql, qr, wave, meqn, mwaves, n, 0);
    } else if (lim_type == 1)
        tvd2(q, ql, qr, meqn, n);
    else if (lim_type == 2 && orc_weno_order == 5)
        weno5_pyweno(q, ql, qr, meqn, n, 3, n - 2);     /* every index whose 5-point stencil is in range */
    else if (lim_type == 2) {                           /* reconstruct.f90:96-114: weno_order 7 .. 17 */
        if (orc_weno_order < 5 || orc_weno_order > 17 || !(orc_weno_order & 1) || mbc < (orc_weno_order + 1) / 2) return -3;
        weno_pyweno_k((orc_weno_order + 1) / 2, q, ql, qr, meqn, n);
    }
    else if (lim_type == 3)
        weno5_legacy(q, ql, qr, meqn, n, mbc);
    else
        return -2;
    int rc;
    if (ndim == 1) rc = orc_rp1_ptr(rp, par, meqn, mwaves, mbc, mx, ql, qr, wave, s, amdq, apdq);
    else rc = orc_rpn2_ptr(rp, par, ixy, meqn, mwaves, mbc, mx, ql, qr, wave, s, amdq, apdq);
    if (rc) return rc;
#define IX(i) ((i) + mbc - 1)
    double cfl = 0.0;
    for (int mw = 0; mw < mwaves; mw++)
        for (int i = 1; i <= mx + 1; i++) {
            double sv = s[mw + mwaves * IX(i)];
            cfl = dmax(dmax(cfl, dtdx[IX(i)] * sv), -dtdx[IX(i - 1)] * sv);
        }
    *cfl_out = cfl;
    /* swap: in-cell Riemann problem between ql(i) and qr(i)  (flux1.f90:166-171) */
    for (int i = 1 - mbc + 1; i <= mx + mbc; i++)
        for (int m = 0; m < meqn; m++) {
            qr[m + meqn * IX(i - 1)] = ql[m + meqn * IX(i)];
            ql[m + meqn * IX(i)] = qr[m + meqn * IX(i)];
        }
    /* flux1.f90:173-178: auxr(:,i-1) = aux(:,i), auxl(:,i) = aux(:,i): both edge states of cell i see its own aux */
    {
        extern const double *orc_aux1d, *orc_auxr1d;
        extern int orc_maux1d;
        double *auxr = NULL;
        if (orc_aux1d && orc_maux1d > 0) {
            auxr = calloc((size_t)orc_maux1d * n, sizeof(double));
            for (int i = 1 - mbc + 1; i <= mx + mbc; i++)
                for (int ma = 0; ma < orc_maux1d; ma++)
                    auxr[ma + orc_maux1d * IX(i - 1)] = orc_aux1d[ma + orc_maux1d * IX(i)];
            orc_auxr1d = auxr;
        }
        if (ndim == 1) rc = orc_rp1_ptr(rp, par, meqn, mwaves, mbc, mx, ql, qr, wave, s, amdq2, apdq2);
        else rc = orc_rpn2_ptr(rp, par, ixy, meqn, mwaves, mbc, mx, ql, qr, wave, s, amdq2, apdq2);
        orc_auxr1d = NULL;
        free(auxr);
    }
    if (rc) return rc;
    for (int i = 1; i <= mx; i++)
        for (int m = 0; m < meqn; m++)
            dq1d[m + meqn * IX(i)] = dq1d[m + meqn * IX(i)] -
                                     dtdx[IX(i)] * (amdq[m + meqn * IX(i + 1)] + apdq[m + meqn * IX(i)] +
                                                    amdq2[m + meqn * IX(i)] + apdq2[m + meqn * IX(i)]);
#undef IX
    return 0;
}

/* sharpclaw2.flux2(q,aux,dt,t,mbc,maxm,mx,my) -> (dq,cfl); dq (meqn, mx+2mbc, my+2mbc) zero-filled on entry */
int orc_sharp_flux2(int rp, const double *par, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, int my, const double *q, double *dq, const double *aux, double dx,
                    double dy, double dt, double *cfl_out)
{
    const int I = mx + 2 * mbc, J = my + 2 * mbc, nmax = (I > J ? I : J);
    double *q1d = calloc((size_t)meqn * nmax, sizeof(double));
    double *dq1d = calloc((size_t)meqn * nmax, sizeof(double));
    double *dtdx = calloc(nmax, sizeof(double));
    double *work = calloc((size_t)meqn * nmax * 6 + (size_t)meqn * mwaves * nmax + (size_t)mwaves * nmax, sizeof(double));
    double cfl = 0.0, cfl1d;
    int rc = 0;
    extern const double *orc_aux1d;
    extern int orc_maux1d;
    double *aux1d = calloc((size_t)nmax * (maux > 0 ? maux : 1), sizeof(double));
    orc_aux1d = aux1d;
    orc_maux1d = maux;
    memset(dq, 0, sizeof(double) * (size_t)meqn * I * J);
#define G(m, i, j) ((m) + (size_t)meqn * (((i) + mbc - 1) + (size_t)I * ((j) + mbc - 1)))
#define A(ma, i, j) aux[((ma)-1) + (size_t)maux * (((i) + mbc - 1) + (size_t)I * ((j) + mbc - 1))]
    for (int j = 0; j <= my + 1 && !rc; j++) {
        for (int i = 1 - mbc; i <= mx + mbc; i++) {
            for (int m = 0; m < meqn; m++) q1d[m + meqn * (i + mbc - 1)] = q[G(m, i, j)];
            dtdx[i + mbc - 1] = (mcapa > 0) ? dt / (dx * A(mcapa, i, j)) : dt / dx;
            for (int ma = 1; ma <= maux; ma++) aux1d[(ma - 1) + maux * (i + mbc - 1)] = A(ma, i, j);
        }
        memset(dq1d, 0, sizeof(double) * (size_t)meqn * nmax);
        rc = flux1(2, rp, par, lim_type, 1, meqn, mwaves, mbc, mx, q1d, dq1d, dtdx, &cfl1d, work);
        cfl = dmax(cfl, cfl1d);
        for (int i = 1; i <= mx; i++)
            for (int m = 0; m < meqn; m++) dq[G(m, i, j)] = dq[G(m, i, j)] + dq1d[m + meqn * (i + mbc - 1)];
    }
    for (int i = 0; i <= mx + 1 && !rc; i++) {
        for (int j = 1 - mbc; j <= my + mbc; j++) {
            for (int m = 0; m < meqn; m++) q1d[m + meqn * (j + mbc - 1)] = q[G(m, i, j)];
            dtdx[j + mbc - 1] = (mcapa > 0) ? dt / (dy * A(mcapa, i, j)) : dt / dy;
            for (int ma = 1; ma <= maux; ma++) aux1d[(ma - 1) + maux * (j + mbc - 1)] = A(ma, i, j);
        }
        memset(dq1d, 0, sizeof(double) * (size_t)meqn * nmax);
        rc = flux1(2, rp, par, lim_type, 2, meqn, mwaves, mbc, my, q1d, dq1d, dtdx, &cfl1d, work);
        cfl = dmax(cfl, cfl1d);
        for (int j = 1; j <= my; j++)
            for (int m = 0; m < meqn; m++) dq[G(m, i, j)] = dq[G(m, i, j)] + dq1d[m + meqn * (j + mbc - 1)];
    }
    *cfl_out = cfl;
    orc_aux1d = NULL;
    free(aux1d);
    free(q1d); free(dq1d); free(dtdx); free(work);
    return rc;
}

/* sharpclaw1.flux1(q,aux,dt,t,ixy,mx,mbc,maxnx) -> (dq,cfl) */
int orc_sharp_flux1(int rp, const double *par, int lim_type, int meqn, int mwaves, int maux, int mcapa,
                    int mbc, int mx, const double *q, double *dq, const double *aux, double dx, double dt,
                    double *cfl_out)
{
    const int n = mx + 2 * mbc;
    double *dtdx = calloc(n, sizeof(double));
    double *work = calloc((size_t)meqn * n * 6 + (size_t)meqn * mwaves * n + (size_t)mwaves * n, sizeof(double));
    for (int i = 0; i < n; i++)
        dtdx[i] = (mcapa > 0) ? dt / (dx * aux[(mcapa - 1) + (size_t)maux * i]) : dt / dx;
    memset(dq, 0, sizeof(double) * (size_t)meqn * n);   /* f2py zero-fills the optional dq1d */
    extern const double *orc_aux1d;
    extern int orc_maux1d;
    orc_aux1d = (maux > 0) ? aux : NULL;     /* 1-D: the aux array is the slice */
    orc_maux1d = maux;
    int rc = flux1(1, rp, par, lim_type, 1, meqn, mwaves, mbc, mx, q, dq, dtdx, cfl_out, work);
    orc_aux1d = NULL;
    free(dtdx); free(work);
    return rc;
}

/* the generic form, any order 5..17 (tests: k = 3 == the hand-written weno5 above) */
void orc_weno_k(int order, int meqn, int n, const double *q, double *ql, double *qr)
{
    weno_pyweno_k((order + 1) / 2, q, ql, qr, meqn, n);
}

void orc_weno5(int variant, int meqn, int n, int mbc, const double *q, double *ql, double *qr)
{
    if (variant == 2) weno5_pyweno(q, ql, qr, meqn, n, 3, n - 2);
    else weno5_legacy(q, ql, qr, meqn, n, mbc);
}
