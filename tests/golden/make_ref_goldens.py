#!/usr/bin/env python
"""
Generates tests/golden/ref_*.npz from the REFERENCE'S OWN FORTRAN (oracle/_ref: flang build of the files
under /root/reference, oracle/Makefile) for the paths the reference ships no golden for:

  ref_step2_unsplit.npz    step2.f  (unsplit, method(3) = 0, 1, 2; rpt2_euler_5wave_rec_loc.f)
  ref_step2ds_capa.npz     step2ds.f with a capacity function (mcapa = 2), ids = 1, 2
  ref_sharp_flux2.npz      SharpClaw flux2.f90, lim_type 2 (PyWENO weno5) and 3 (legacy weno5)
  ref_step2_unsplit_capa.npz  step2.f with a capacity function (mcapa = 2), method(3) = 0, 1, 2
  ref_sharp_tvd2.npz       SharpClaw flux2.f90 with lim_type 1 (tvd2), mthlim 1..5
  ref_sharp_weno_orders.npz  SharpClaw flux2.f90 with lim_type 2 and weno_order 7, 9, ..., 17 (weno.f90)
  ref_flux2fw.npz          flux2fw.f (the classic2fw link of step2ds.f / step2.f): dim-split x / y, unsplit trans 0/1/2
  ref_recon_wave.npz       tvd2_wave / weno5_wave / weno5_fwave of 1d/sharpclaw/reconstruct.f90 (char_decomp = 1)
  ref_sphere_setup.npz     the shallow-sphere app's own setaux.f / qinit.f / src2.f / qcor.f (40 x 20 grid)

Only inputs' SEEDS and the outputs are stored (inputs are regenerated from the seed by the tests).  Run in the
build container (needs oracle/_ref, i.e. the reference tree): python tests/golden/make_ref_goldens.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O          # noqa: E402


def euler_state(seed, shape):
    rng = np.random.default_rng(seed)
    q = np.empty((5,) + shape, order="F")
    rho = 0.5 + rng.random(shape)
    u = 1.5 * (rng.random(shape) - 0.5)
    v = 1.5 * (rng.random(shape) - 0.5)
    p = 0.5 + rng.random(shape)
    q[0] = rho
    q[1] = rho * u
    q[2] = rho * v
    q[3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    q[4] = rng.random(shape)
    return q


def euler_state_mild(seed, shape):
    """jumps of ~10 % between neighbours: rough enough to exercise every nonlinear weight, mild enough that the
    reconstructions of order 13 and 15 keep density and pressure positive (on euler_state they do not, and the
    reference's own result then holds NaNs)"""
    rng = np.random.default_rng(seed)
    q = np.empty((5,) + shape, order="F")
    rho = 1.0 + 0.2 * rng.random(shape)
    u = 0.3 * (rng.random(shape) - 0.5)
    v = 0.3 * (rng.random(shape) - 0.5)
    p = 1.0 + 0.2 * rng.random(shape)
    q[0] = rho
    q[1] = rho * u
    q[2] = rho * v
    q[3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    q[4] = rng.random(shape)
    return q


def capa_field(seed, shape):
    return np.asfortranarray(0.5 + np.random.default_rng(seed + 1000).random((2,) + shape))


PAR = [1.4, 0.4]
MTH = [4, 4, 4, 4, 2]


def recon_wave_inputs(kind, lim):
    """q, wave, s for the wave-based reconstructions: a smooth profile with a jump, waves that partly vanish (the
    wnorm2 <= 1e-14 and wnorm2 == 0 branches), speeds of both signs"""
    rng = np.random.default_rng(900 + 10 * kind + lim)
    meqn, mwaves, n = 3, 3, 46
    x = np.arange(n) / float(n)
    q = np.asfortranarray(np.stack([1.0 + 0.3 * np.sin(6.0 * x) + (x > 0.6), 0.2 * rng.random(n) - 0.1, 2.0 + 0.1 * rng.standard_normal(n)]))
    wave = np.asfortranarray(0.3 * rng.standard_normal((meqn, mwaves, n)))
    wave[:, 1, 10:14] = 0.0                       # a family without a wave
    wave[:, 2, 20:23] *= 1e-9                     # |wave|^2 below the 1e-14 switch of weno5_wave
    s = np.asfortranarray(rng.standard_normal((mwaves, n)))
    s = np.where(np.abs(s) < 0.05, 0.05, s)       # weno5_fwave divides by s
    return q, wave, np.asfortranarray(s)


def main():
    # ---- char_decomp = 1: the wave-based reconstructions, FIRST (shim comment at sc_recon_wave)
    srw = O.RefSharp2DEuler()
    out = {}
    for kind, lims in ((1, (1, 2, 3, 4, 5)), (2, (0,)), (3, (0,))):
        for lim in lims:
            q, wave, s = recon_wave_inputs(kind, lim)
            ql, qr = srw.recon_wave(kind, q, wave, s, [max(lim, 1)] * 3)
            out["ql_kind%d_lim%d" % (kind, lim)] = ql
            out["qr_kind%d_lim%d" % (kind, lim)] = qr
    np.savez_compressed(os.path.join(HERE, "ref_recon_wave.npz"), **out)

    ref = O.RefEuler2D()
    mx, my, mbc = 37, 26, 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    dx, dy, dt = 1.0 / mx, 0.8 / my, 0.004
    out = {}
    for trans in (0, 1, 2):
        q0 = euler_state(10 + trans, shape)
        qn = q0.copy("F")
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        _, cfl = ref.step2(O.RP_EULER5_2D, PAR, max(mx, my), mbc, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method, MTH)
        out["q_trans%d" % trans] = qn
        out["cfl_trans%d" % trans] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_step2_unsplit.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    out = {}
    method = np.array([1, 2, -1, 0, 0, 2, 2], dtype=np.int32)
    for ids in (1, 2):
        q0 = euler_state(20 + ids, shape)
        aux = capa_field(20 + ids, shape)
        qn = q0.copy("F")
        _, cfl = ref.step2ds(O.RP_EULER5_2D, PAR, max(mx, my), mbc, mx, my, q0.copy("F"), qn, aux, dx, dy, dt, method, MTH, ids)
        out["q_ids%d" % ids] = qn
        out["cfl_ids%d" % ids] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_step2ds_capa.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    # unsplit algorithm WITH a capacity function (the annulus / sphere configuration): step2.f:145-152,227-234
    out = {}
    for trans in (0, 1, 2):
        q0 = euler_state(40 + trans, shape)
        aux = capa_field(40 + trans, shape)
        qn = q0.copy("F")
        method = np.array([1, 2, trans, 0, 0, 2, 2], dtype=np.int32)
        _, cfl = ref.step2(O.RP_EULER5_2D, PAR, max(mx, my), mbc, mx, my, q0.copy("F"), qn, aux, dx, dy, dt, method, MTH)
        out["q_trans%d" % trans] = qn
        out["cfl_trans%d" % trans] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_step2_unsplit_capa.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    # the f-wave form of the correction terms (flux2fw.f:145-152: dsign(1,s) in place of |s|), linked with the vendored
    # Euler solver: pins the ARITHMETIC of flux2fw.f (the Euler solver returns waves, so the numbers are no physics)
    reffw = O.RefEuler2D(fwave=True)
    out = {}
    method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
    for ids in (1, 2):
        q0 = euler_state(70 + ids, shape)
        qn = q0.copy("F")
        _, cfl = reffw.step2ds(O.RP_EULER5_2D, PAR, max(mx, my), mbc, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method,
                               MTH, ids, fwave=True)
        out["q_ids%d" % ids] = qn
        out["cfl_ids%d" % ids] = cfl
    for trans in (0, 1, 2):
        q0 = euler_state(80 + trans, shape)
        qn = q0.copy("F")
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        _, cfl = reffw.step2(O.RP_EULER5_2D, PAR, max(mx, my), mbc, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method, MTH,
                             fwave=True)
        out["q_trans%d" % trans] = qn
        out["cfl_trans%d" % trans] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_flux2fw.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    # shallow water on the sphere: the app's own data generators and source / correction terms
    sp = O.RefSphereProblem()
    smx, smy, g = 40, 20, 2
    sdx, sdy = 4.0 / smx, 2.0 / smy
    aux = sp.sphere_setaux(g, smx, smy, -3.0, -1.0, sdx, sdy)
    q0 = sp.sphere_qinit(g, smx, smy, -3.0, -1.0, sdx, sdy)
    qi = np.array(q0[:, g:-g, g:-g], order="F")
    qs = qi.copy(order="F")
    sp.sphere_src2(qs, np.array(aux[:, g:-g, g:-g], order="F"), -3.0, -1.0, sdx, sdy, 0.01)
    qc = np.zeros((2, 4, 8))
    for ixy in (1, 2):
        for k in range(8):           # qcor at 8 cells of one slice (Fortran indices 3 .. 10)
            if ixy == 1:
                a1, q1 = np.array(aux[:, :, 7], order="F"), np.array(q0[:, :, 7], order="F")
            else:
                a1, q1 = np.array(aux[:, 9, :], order="F"), np.array(q0[:, 9, :], order="F")
            qc[ixy - 1, :, k] = sp.qcor(ixy, 3 + k, a1, q1, g, 11489.57219, sdx, sdy)
    np.savez_compressed(os.path.join(HERE, "ref_sphere_setup.npz"), mx=smx, my=smy, aux=aux, q0=qi, q_src2=qs,
                        dt_src2=0.01, qcor=qc)

    sref = O.RefSharp2DEuler()
    mbc = 3
    shape = (mx + 2 * mbc, my + 2 * mbc)
    out = {}
    for lim in (2, 3):
        q0 = euler_state(30 + lim, shape)
        dq, cfl = sref.sharp_flux2(O.RP_EULER5_2D, PAR, lim, 5, 0, mbc, mx, my, q0, None, dx, dy, dt)
        out["dq_lim%d" % lim] = dq[:, mbc:-mbc, mbc:-mbc]
        out["cfl_lim%d" % lim] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_sharp_flux2.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    # lim_type = 1: tvd2 (reconstruct.f90:568-625), every limiter id; a generic state (no constant patch, see
    # oracle/sharpclaw_oracle.c).  The first interior row / column of the reference result depends on an
    # uninitialised variable of the Fortran and is NOT compared by the tests.
    out = {}
    for lim in (1, 2, 3, 4, 5):
        q0 = euler_state(50 + lim, shape)
        dq, cfl = sref.sharp_flux2(O.RP_EULER5_2D, PAR, 1, 5, 0, mbc, mx, my, q0, None, dx, dy, dt, mthlim=[lim] * 5)
        out["dq_mth%d" % lim] = dq[:, mbc:-mbc, mbc:-mbc]
    np.savez_compressed(os.path.join(HERE, "ref_sharp_tvd2.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)

    # lim_type = 2 with weno_order 7 .. 17 (weno.f90: weno7 ... weno17), mbc = (weno_order+1)/2
    out = {}
    for order in (7, 9, 11, 13, 15, 17):
        g = (order + 1) // 2
        q0 = euler_state_mild(60 + order, (mx + 2 * g, my + 2 * g))
        dq, cfl = sref.sharp_flux2(O.RP_EULER5_2D, PAR, 2, 5, 0, g, mx, my, q0, None, dx, dy, dt, weno_order=order)
        assert np.isfinite(dq).all()
        out["dq_order%d" % order] = dq[:, g:-g, g:-g]
        out["cfl_order%d" % order] = cfl
    np.savez_compressed(os.path.join(HERE, "ref_sharp_weno_orders.npz"), mx=mx, my=my, dx=dx, dy=dy, dt=dt, **out)
    print("written:", [f for f in sorted(os.listdir(HERE)) if f.startswith("ref_")])


if __name__ == "__main__":
    main()
