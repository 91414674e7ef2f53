/*
 * oracle/sphere_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).
 *
 * C restatement of the shallow-water-on-the-sphere application's own Fortran (the reference keeps these next to
 * the app: test/shallow_sphere/ == apps/shallow-sphere/):
 *     mapc2p   test/shallow_sphere/mapc2p.f:1-76
 *     setaux   test/shallow_sphere/setaux.f:1-218
 *     qinit    test/shallow_sphere/qinit.f:1-107
 *     src2     test/shallow_sphere/src2.f:1-147
 * Operation order follows the Fortran line by line.  Pinned against the reference's own build of these four files
 * (oracle/_ref/libref_sphere_problem.so, tests/test_oracle_vs_ref.py) and, through the whole run, against
 * test/swsphere_height.  (The libm calls -- acos, asin, atan, tan, cos, sin, pow -- may differ by an ulp between
 * flang's runtime and glibc; the comparison allows for that.)
 * Arrays are Fortran ordered, component fastest: aux(maux, 1-mbc:mx+mbc, 1-mbc:my+mbc), q likewise.
 */
#include <math.h>
#include <stdlib.h>

static inline double dmax(double a, double b) { return a > b ? a : b; }

/* mapc2p.f:1-76 */
void orc_sphere_mapc2p(double x1, double y1, double *xp_, double *yp_, double *zp_, double Rsphere)
{
    const double r1 = Rsphere;
    double xc = x1, yc = y1, sgnz;
    if (xc >= 1.0) xc = xc - 4.0;
    if (xc < -3.0) xc = xc + 4.0;
    if (yc >= 1.0) { yc = 2.0 - yc; xc = -2.0 - xc; }
    if (yc < -1.0) { yc = -2.0 - yc; xc = -2.0 - xc; }
    if (xc < -1.0) { xc = -2.0 - xc; sgnz = -1.0; } else sgnz = 1.0;
    const double sgnxc = copysign(1.0, xc), sgnyc = copysign(1.0, yc);
    const double xc1 = fabs(xc), yc1 = fabs(yc);
    const double d = dmax(dmax(xc1, yc1), 1.e-10);
    const double DD = r1 * d * (2.0 - d) / sqrt(2.0);
    const double R = r1;
    const double center = DD - sqrt(dmax(R * R - DD * DD, 0.0));
    double xp = DD / d * xc1;
    double yp = DD / d * yc1;
    if (yc1 > xc1) yp = center + sqrt(dmax(R * R - xp * xp, 0.0));
    else xp = center + sqrt(dmax(R * R - yp * yp, 0.0));
    const double zp = sqrt(dmax(r1 * r1 - (xp * xp + yp * yp), 0.0));
    *xp_ = xp * sgnxc;
    *yp_ = yp * sgnyc;
    *zp_ = zp * sgnz;
}

/* setaux.f:48-216; aux(16, mx+2mbc, my+2mbc) */
int orc_sphere_setaux(int mbc, int mx, int my, double xlower, double ylower, double dxc, double dyc, double *aux,
                      double Rsphere)
{
    const int maux = 16;
    const double pi = 4.0 * atan(1.0);
    const int ni = mx + 2 * mbc + 1, nj = my + 2 * mbc + 1;     /* corners i = 1-mbc .. mx+mbc+1 */
    double *xp = malloc(sizeof(double) * ni * nj), *yp = malloc(sizeof(double) * ni * nj);
    double *zp = malloc(sizeof(double) * ni * nj), *theta = calloc((size_t)ni * nj, sizeof(double));
    double *phi = malloc(sizeof(double) * ni * nj);
    if (!xp || !yp || !zp || !theta || !phi) return -1;
#define C(arr, i, j) arr[((i) + mbc - 1) + (size_t)ni * ((j) + mbc - 1)]
#define AUX(ma, i, j) aux[((ma)-1) + (size_t)maux * (((i) + mbc - 1) + (size_t)(mx + 2 * mbc) * ((j) + mbc - 1))]
    for (int j = 1 - mbc; j <= my + mbc + 1; j++)
        for (int i = 1 - mbc; i <= mx + mbc + 1; i++) {
            const double xc = xlower + (i - 1.0) * dxc, yc = ylower + (j - 1.0) * dyc;
            orc_sphere_mapc2p(xc, yc, &C(xp, i, j), &C(yp, i, j), &C(zp, i, j), Rsphere);
            const double r = sqrt(C(xp, i, j) * C(xp, i, j) + C(yp, i, j) * C(yp, i, j));
            if (r > 1.e-4) C(theta, i, j) = acos(C(xp, i, j) / r);
            else if (C(yp, i, j) > 0.0) C(theta, i, j) = 0.0;
            if (C(yp, i, j) < 0.0) C(theta, i, j) = -C(theta, i, j);
            if (C(zp, i, j) > 0.0) C(phi, i, j) = pi / 2.0 - acos(r / Rsphere);
            else C(phi, i, j) = pi / 2.0 + acos(r / Rsphere);
        }
    for (int j = 1 - mbc; j <= my + mbc; j++)
        for (int i = 1 - mbc; i <= mx + mbc; i++) {
            double etx, ety, etz, erx, ery, erz, enx, eny, enz, ennorm;
            /* left edge */
            etx = C(xp, i, j + 1) - C(xp, i, j);
            ety = C(yp, i, j + 1) - C(yp, i, j);
            etz = C(zp, i, j + 1) - C(zp, i, j);
            AUX(5, i, j) = etx; AUX(6, i, j) = ety; AUX(7, i, j) = etz;
            erx = 0.5 * (C(xp, i, j) + C(xp, i, j + 1));
            ery = 0.5 * (C(yp, i, j) + C(yp, i, j + 1));
            erz = 0.5 * (C(zp, i, j) + C(zp, i, j + 1));
            enx = ety * erz - etz * ery;
            eny = etz * erx - etx * erz;
            enz = etx * ery - ety * erx;
            ennorm = sqrt(enx * enx + eny * eny + enz * enz);
            AUX(2, i, j) = enx / ennorm; AUX(3, i, j) = eny / ennorm; AUX(4, i, j) = enz / ennorm;
            /* bottom edge */
            etx = C(xp, i + 1, j) - C(xp, i, j);
            ety = C(yp, i + 1, j) - C(yp, i, j);
            etz = C(zp, i + 1, j) - C(zp, i, j);
            AUX(11, i, j) = etx; AUX(12, i, j) = ety; AUX(13, i, j) = etz;
            erx = 0.5 * (C(xp, i, j) + C(xp, i + 1, j));
            ery = 0.5 * (C(yp, i, j) + C(yp, i + 1, j));
            erz = 0.5 * (C(zp, i, j) + C(zp, i + 1, j));
            enx = ery * etz - erz * ety;
            eny = erz * etx - erx * etz;
            enz = erx * ety - ery * etx;
            ennorm = sqrt(enx * enx + eny * eny + enz * enz);
            AUX(8, i, j) = enx / ennorm; AUX(9, i, j) = eny / ennorm; AUX(10, i, j) = enz / ennorm;
            /* radial direction at the cell centre; setaux.f:150-151 writes (i-0.5) with a REAL*4 literal, which is
             * exact (0.5) */
            double xpm, ypm, zpm;
            orc_sphere_mapc2p(xlower + (i - 0.5) * dxc, ylower + (j - 0.5) * dyc, &xpm, &ypm, &zpm, Rsphere);
            AUX(14, i, j) = xpm; AUX(15, i, j) = ypm; AUX(16, i, j) = zpm;
            /* area of the cell from two spherical triangles, setaux.f:166-211 */
#define BETA(i1, j1, i2, j2) (sin(C(phi, i1, j1)) * sin(C(phi, i2, j2)) * cos(C(theta, i1, j1) - C(theta, i2, j2)) + \
                              cos(C(phi, i1, j1)) * cos(C(phi, i2, j2)))
            const double beta12 = BETA(i, j, i + 1, j);
            const double beta23 = BETA(i, j + 1, i + 1, j);
            const double beta13 = BETA(i, j + 1, i, j);
            const double beta24 = BETA(i + 1, j + 1, i + 1, j);
            const double beta34 = BETA(i + 1, j + 1, i, j + 1);
#undef BETA
            const double d12 = Rsphere * acos(beta12), d23 = Rsphere * acos(beta23), d13 = Rsphere * acos(beta13);
            const double d24 = Rsphere * acos(beta24), d34 = Rsphere * acos(beta34);
            const double s123 = 0.5 * (d12 + d23 + d13), s234 = 0.5 * (d23 + d34 + d24);
            double t123 = tan(s123 / 2.0) * tan((s123 - d12) / 2.0) * tan((s123 - d23) / 2.0) * tan((s123 - d13) / 2.0);
            t123 = dmax(t123, 0.0);
            const double E123 = 4.0 * atan(sqrt(t123));
            double t234 = tan(s234 / 2.0) * tan((s234 - d23) / 2.0) * tan((s234 - d34) / 2.0) * tan((s234 - d24) / 2.0);
            t234 = dmax(t234, 0.0);
            const double E234 = 4.0 * atan(sqrt(t234));
            const double area = (E123 + E234);
            AUX(1, i, j) = area / (dxc * dyc);
        }
    free(xp); free(yp); free(zp); free(theta); free(phi);
    return 0;
#undef C
}

/* qinit.f:30-105: 4-Rossby-Haurwitz wave; q(4, mx+2mbc, my+2mbc), interior cells only are set */
int orc_sphere_qinit(int mbc, int mx, int my, double xlower, double ylower, double dx, double dy, double *q,
                     double Rsphere)
{
    const int meqn = 4;
    const double pi = 4.0 * atan(1.0);
    const double a = 6.37122e6, K = 7.848e-6, Omega = 7.292e-5, G = 9.80616, t0 = 86400.0, h0 = 8.e3, R = 4.0;
#define Q(m, i, j) q[((m)-1) + (size_t)meqn * (((i) + mbc - 1) + (size_t)(mx + 2 * mbc) * ((j) + mbc - 1))]
    for (int i = 1; i <= mx; i++) {
        const double xc = xlower + (i - 0.5) * dx;
        for (int j = 1; j <= my; j++) {
            const double yc = ylower + (j - 0.5) * dy;
            double xp, yp, zp, theta = 0.0, phi;
            orc_sphere_mapc2p(xc, yc, &xp, &yp, &zp, Rsphere);
            const double rad = dmax(sqrt(xp * xp + yp * yp), 1.e-6);
            if (xp > 0.0 && yp > 0.0) theta = asin(yp / rad);
            else if (xp < 0.0 && yp > 0.0) theta = pi - asin(yp / rad);
            else if (xp < 0.0 && yp < 0.0) theta = -pi + asin(-yp / rad);
            else if (xp > 0.0 && yp < 0.0) theta = -asin(-yp / rad);
            if (zp > 0.0) phi = asin(zp / Rsphere);
            else phi = -asin(-zp / Rsphere);
            xp = theta;
            yp = phi;
            const double cy = cos(yp), sy = sin(yp);
            const double bigA = 0.5 * K * (2.0 * Omega + K) * pow(cy, 2.0) +
                                0.25 * K * K * pow(cy, 2.0 * R) *
                                    ((1.0 * R + 1.0) * pow(cy, 2.0) + (2.0 * R * R - 1.0 * R - 2.0) -
                                     2.0 * R * R * pow(cy, -2.0));
            const double bigB = (2.0 * (Omega + K) * K) / ((1.0 * R + 1.0) * (1.0 * R + 2.0)) * pow(cy, R) *
                                ((1.0 * R * R + 2.0 * R + 2.0) - ((1.0 * R + 1.0) * (1.0 * R + 1.0)) * (cy * cy));
            const double bigC = 0.25 * K * K * pow(cy, 2 * R) * ((1.0 * R + 1.0) * (cy * cy) - (1.0 * R + 2.0));
            const double Uin1 = (K * cy + K * pow(cy, R - 1.) * (R * pow(sy, 2.) - pow(cy, 2.)) * cos(R * xp)) * t0;
            const double Uin2 = (-K * R * pow(cy, R - 1.) * sy * sin(R * xp)) * t0;
            const double Uout1 = (-sin(xp) * Uin1 - sy * cos(xp) * Uin2);
            const double Uout2 = (cos(xp) * Uin1 - sy * sin(xp) * Uin2);
            const double Uout3 = cy * Uin2;
            Q(1, i, j) = h0 / a + (a / G) * (bigA + bigB * cos(R * xp) + bigC * cos(2.0 * R * xp));
            Q(2, i, j) = Q(1, i, j) * Uout1;
            Q(3, i, j) = Q(1, i, j) * Uout2;
            Q(4, i, j) = Q(1, i, j) * Uout3;
        }
    }
    return 0;
#undef Q
}

/* src2.f:43-146: Coriolis force (4-stage RK) between two projections onto the tangent plane.
 * q(meqn, 1:mx, 1:my) and aux(maux, 1:mx, 1:my) WITHOUT ghost cells (the Python wrapper passes state.q/state.aux,
 * shallow_4_Rossby_Haurwitz_wave.py:27-49). */
int orc_sphere_src2(int meqn, int mx, int my, double xlower, double ylower, double dx, double dy, double *q, int maux,
                    const double *aux, double dt, double Rsphere)
{
    const double df = (double)12.600576f;     /* src2.f:39: df=12.600576e0, a REAL*4 literal */
#define Q(m, i, j) q[((m)-1) + (size_t)meqn * (((i)-1) + (size_t)mx * ((j)-1))]
#undef AUX
#define AUX(ma, i, j) aux[((ma)-1) + (size_t)maux * (((i)-1) + (size_t)mx * ((j)-1))]
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 1; i <= mx; i++)
            for (int j = 1; j <= my; j++) {
                const double erx = AUX(14, i, j), ery = AUX(15, i, j), erz = AUX(16, i, j);
                const double qn = erx * Q(2, i, j) + ery * Q(3, i, j) + erz * Q(4, i, j);
                Q(2, i, j) = Q(2, i, j) - qn * erx;
                Q(3, i, j) = Q(3, i, j) - qn * ery;
                Q(4, i, j) = Q(4, i, j) - qn * erz;
            }
        if (pass == 1) break;
        for (int i = 1; i <= mx; i++) {
            const double xc = xlower + (i - 0.5) * dx;
            for (int j = 1; j <= my; j++) {
                const double yc = ylower + (j - 0.5) * dy;
                double erx, ery, erz;
                orc_sphere_mapc2p(xc, yc, &erx, &ery, &erz, Rsphere);
                const double fcor = df * erz;
                double RK[5][4], hu, hv, hw;
                hu = Q(2, i, j); hv = Q(3, i, j); hw = Q(4, i, j);
                RK[1][1] = fcor * dt * (erz * hv - ery * hw);
                RK[1][2] = dt * fcor * (erx * hw - erz * hu);
                RK[1][3] = dt * fcor * (ery * hu - erx * hv);
                for (int st = 2; st <= 4; st++) {
                    hu = Q(2, i, j) + 0.5 * RK[st - 1][1];
                    hv = Q(3, i, j) + 0.5 * RK[st - 1][2];
                    hw = Q(4, i, j) + 0.5 * RK[st - 1][3];
                    RK[st][1] = fcor * dt * (erz * hv - ery * hw);
                    RK[st][2] = dt * fcor * (erx * hw - erz * hu);
                    RK[st][3] = dt * fcor * (ery * hu - erx * hv);
                }
                for (int m = 2; m <= meqn; m++)
                    Q(m, i, j) = Q(m, i, j) + (RK[1][m - 1] + 2.0 * RK[2][m - 1] + 2.0 * RK[3][m - 1] + RK[4][m - 1]) / 6.0;
            }
        }
    }
    return 0;
#undef Q
#undef AUX
}
