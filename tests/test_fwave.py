"""
The f-wave path (flux2fw.f:151-152, step1fw.f:135-141) with the two restated f-wave solvers (third-party, absent from
the reference tree, no golden: PARITY UNPINNED at the solver boundary -- oracle/classic_oracle.c: rp_fwave_normal).
What CAN be held without the third-party sources:
  * CPU: for a linear stress law in a uniform medium the p-system IS acoustics (p = -K eps, (u,v) = m/rho): the f-wave
    path with the restated solvers must give the wave path's result with the reference-pinned acoustics solver, to
    rounding -- normal solver, transverse solver, limiter on f-waves and the dsign(1,s) correction all take part;
  * GPU (-m gpu): HIP == oracle bit for bit (linear law, heterogeneous medium), to 1e-13 for the exponential law (the
    device's exp() and glibc's differ by an ulp).
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O

RHO, K = 1.7, 3.1
CC = np.sqrt(K / RHO)
ZZ = RHO * CC


def checkerboard_aux(shape, lin, rng, hetero=True):
    aux = np.zeros((4,) + shape, order="F")
    i = np.arange(shape[0])[:, None]
    j = np.arange(shape[-1])[None, :] if len(shape) == 2 else 0
    cell = ((i // 5 + j // 4) % 2 == 0) if len(shape) == 2 else ((np.arange(shape[0]) // 5) % 2 == 0)
    aux[0] = np.where(cell, 1.0, 4.0) if hetero else RHO
    aux[1] = np.where(cell, 1.0, 4.0) if hetero else K
    aux[2] = lin
    return aux


@pytest.mark.parametrize("trans", [0, 1, 2])
def test_oracle_psystem_linear_uniform_equals_acoustics(coracle, trans):
    rng = np.random.default_rng(1)
    mx, my, mbc = 40, 31, 2
    shape = (mx + 4, my + 4)
    p, u, v = rng.standard_normal(shape), rng.standard_normal(shape), rng.standard_normal(shape)
    qa = np.asfortranarray(np.stack([p, u, v]))
    qp = np.asfortranarray(np.stack([-p / K, RHO * u, RHO * v]))
    aux = checkerboard_aux(shape, 1.0, rng, hetero=False)
    aux[3] = qp[0]
    dx, dy, dt = 0.05, 0.06, 0.01
    mth = np.array([4, 2], dtype=np.int32)
    method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
    ra = qa.copy("F")
    coracle.step2(O.RP_ACOUSTICS_2D, [RHO, K, CC, ZZ], max(mx, my), mbc, mx, my, qa.copy("F"), ra, None, dx, dy, dt, method, mth)
    mp = method.copy()
    mp[6] = 4
    rp = qp.copy("F")
    coracle.step2(O.RP_PSYSTEM_FWAVE_2D, [0.0], max(mx, my), mbc, mx, my, qp.copy("F"), rp, aux, dx, dy, dt, mp, mth, fwave=True)
    inner = (slice(None), slice(2, -2), slice(2, -2))
    back = np.stack([-K * rp[0], rp[1] / RHO, rp[2] / RHO])
    assert np.abs(back[inner] - ra[inner]).max() < 2e-14
    for ids in (1, 2):            # and the dimension-split sweeps
        md, mdp = method.copy(), mp.copy()
        md[2] = mdp[2] = -1
        ra = qa.copy("F")
        coracle.step2ds(O.RP_ACOUSTICS_2D, [RHO, K, CC, ZZ], max(mx, my), mbc, mx, my, qa.copy("F"), ra, None, dx, dy, dt, md, mth, ids)
        rp = qp.copy("F")
        coracle.step2ds(O.RP_PSYSTEM_FWAVE_2D, [0.0], max(mx, my), mbc, mx, my, qp.copy("F"), rp, aux, dx, dy, dt, mdp, mth, ids, fwave=True)
        back = np.stack([-K * rp[0], rp[1] / RHO, rp[2] / RHO])
        assert np.abs(back - ra).max() < 2e-14


def test_oracle_step1fw_linear_uniform_equals_step1_acoustics(coracle):
    rng = np.random.default_rng(2)
    mx = 50
    q = np.asfortranarray(rng.standard_normal((2, mx + 4)))
    aux = np.zeros((3, mx + 4), order="F")
    aux[0], aux[1], aux[2] = RHO, K, 1.0
    qa = np.asfortranarray(np.stack([-K * q[0], q[1] / RHO]))
    mth = np.array([4, 3], dtype=np.int32)
    r1 = q.copy("F")
    coracle.step1(O.RP_ELASTICITY_FWAVE_1D, [0.0], 2, mx, r1, aux, 0.1, 0.02, np.array([1, 2, 0, 0, 0, 0, 3], dtype=np.int32),
                  mth, fwave=True)
    r2 = qa.copy("F")
    coracle.step1(O.RP_ACOUSTICS_1D, [RHO, K, CC, ZZ], 2, mx, r2, None, 0.1, 0.02, np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32), mth)
    assert np.abs(np.stack([-K * r1[0], r1[1] / RHO])[:, 2:-2] - r2[:, 2:-2]).max() < 2e-14


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("lin", [1.0, 2.0])
@pytest.mark.parametrize("trans", [-1, 0, 1, 2])
@pytest.mark.parametrize("mx,my", [(23, 17), (130, 66)])
def test_hip_psystem_fwave(coracle, lin, trans, mx, my):
    """classic2fw (flux2fw.f) with rpn2/rpt2_psystem on a checkerboard medium: dim-split sweeps (trans = -1) and
    the unsplit step with method(3) = 0, 1, 2"""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx + 3 * my + trans)
    shape = (mx + 4, my + 4)
    aux = checkerboard_aux(shape, lin, rng)
    q0 = np.asfortranarray(0.3 * rng.standard_normal((3,) + shape))
    aux[3] = q0[0]
    dx, dy = 1.0 / mx, 0.9 / my
    dt = 0.3 * min(dx, dy) / 2.0
    mth = np.array([2, 4], dtype=np.int32)
    method = np.array([1, 2, trans, 0, 0, 0, 4], dtype=np.int32)
    par = np.zeros(8)
    cfl = C.c_double()
    tol = 0.0 if lin == 1.0 else 1e-13
    if trans < 0:
        for ids in (1, 2):
            ref = q0.copy("F")
            _, cfl_ref = coracle.step2ds(O.RP_PSYSTEM_FWAVE_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy, dt,
                                         method, mth, ids, fwave=True)
            out = q0.copy("F")
            L.check(L.lib().pcl_step2ds(O.RP_PSYSTEM_FWAVE_2D, L.d(par), 1, 3, 2, 4, 2, mx, my, L.d(q0), L.d(out), L.d(aux),
                                        dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
            assert np.abs(out - ref).max() <= tol and abs(cfl.value - cfl_ref) <= tol
    else:
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2(O.RP_PSYSTEM_FWAVE_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy, dt,
                                   method, mth, fwave=True)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2(O.RP_PSYSTEM_FWAVE_2D, L.d(par), 1, 3, 2, 4, 2, mx, my, L.d(q0), L.d(out), L.d(aux),
                                  dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        inner = (slice(None), slice(2, -2), slice(2, -2))
        assert np.abs(out[inner] - ref[inner]).max() <= tol and abs(cfl.value - cfl_ref) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("lin", [1.0, 0.0])
@pytest.mark.parametrize("mx", [7, 60, 61, 777])
def test_hip_step1fw_elasticity(coracle, lin, mx):
    """classic1fw.step1 (step1fw.f) with rp1_nonlinear_elasticity_fwave on the stegoton's layered medium"""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx)
    aux = np.asfortranarray(checkerboard_aux((mx + 4,), lin, rng)[:3])
    q = np.asfortranarray(0.2 * rng.standard_normal((2, mx + 4)))
    mth = np.array([4, 1], dtype=np.int32)
    method = np.array([1, 2, 0, 0, 0, 0, 3], dtype=np.int32)
    ref = q.copy("F")
    _, cfl_ref = coracle.step1(O.RP_ELASTICITY_FWAVE_1D, [0.0], 2, mx, ref, aux, 1.0 / mx, 0.2 / mx, method, mth, fwave=True)
    out = q.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1fw(O.RP_ELASTICITY_FWAVE_1D, L.d(np.zeros(8)), 2, 2, 3, 2, mx, L.d(out), L.d(aux), 1.0 / mx,
                                0.2 / mx, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    tol = 0.0 if lin == 1.0 else 1e-13
    assert np.abs(out[:, 2:-2] - ref[:, 2:-2]).max() <= tol and abs(cfl.value - cfl_ref) <= tol


@pytest.mark.gpu
def test_fwave_flag_must_match_the_solver():
    from pyclaw_amd import _lib as L
    q = np.zeros((3, 12, 12), order="F")
    aux = np.ones((4, 12, 12), order="F")
    cfl = C.c_double()
    method = np.array([1, 2, -1, 0, 0, 0, 4], dtype=np.int32)
    mth = np.array([1, 1], dtype=np.int32)
    rc = L.lib().pcl_step2ds(O.RP_PSYSTEM_FWAVE_2D, L.d(np.zeros(8)), 0, 3, 2, 4, 2, 8, 8, L.d(q), L.d(q.copy("F")), L.d(aux),
                             0.1, 0.1, 0.01, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), 1)
    assert rc == L.EINVAL and b"f-waves" in L.lib().pcl_last_error()
    rc = L.lib().pcl_step2ds(O.RP_ACOUSTICS_2D, L.d(np.array([1., 1., 1., 1., 0, 0, 0, 0])), 1, 3, 2, 0, 2, 8, 8, L.d(q),
                             L.d(q.copy("F")), None, 0.1, 0.1, 0.01, L.i(np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)),
                             L.i(mth), C.cast(C.byref(cfl), L.dp), 1)
    assert rc == L.EINVAL
