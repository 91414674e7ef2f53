"""
GPU: SharpClaw (WENO5 + SSP Runge-Kutta) on the device against the oracle (C restatement, itself
bit-identical to the flang build of the reference's flux1/flux2/weno modules) and the reference's
golden test/ac_sc_solution.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from apps import problems
from oracle import driver as D
from oracle import oracle as O


def make(L, ndim, n, rp, meqn, mwaves, par, lim, d, math=0):
    cfg = L.Config()
    cfg.ndim = ndim
    for k in range(ndim):
        cfg.n[k] = n[k]
        cfg.d[k] = d[k]
    cfg.mbc = 3
    cfg.meqn, cfg.mwaves, cfg.rp = meqn, mwaves, rp
    cfg.method[1] = 2
    for k, v in enumerate(par):
        cfg.rp_params[k] = v
    cfg.kind = 1
    cfg.lim_type = lim
    cfg.math = math
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    return h


def euler_q(rng, shape):
    q = np.empty((5,) + shape, order="F")
    rho = 1 + 0.3 * rng.random(shape)
    u = 0.5 * (rng.random(shape) - .5)
    v = 0.5 * (rng.random(shape) - .5)
    p = 1 + 0.3 * rng.random(shape)
    q[0] = rho
    q[1] = rho * u
    q[2] = rho * v
    q[3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    q[4] = rng.random(shape)
    return q


@pytest.mark.parametrize("mx,my", [(1, 1), (9, 5), (58, 58), (59, 57), (120, 75)])
@pytest.mark.parametrize("lim", [2, 3])
def test_flux2_euler_bitexact(coracle, mx, my, lim):
    """device dq == sharpclaw2.flux2 of the oracle, interior cells, bit for bit"""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx * 100 + my)
    mbc = 3
    q = euler_q(rng, (mx + 2 * mbc, my + 2 * mbc))
    par = [1.4, 0.4]
    dx, dy, dt = 1.0 / mx, 0.8 / my, 0.02
    ref, cfl_ref = coracle.sharp_flux2(O.RP_EULER5_2D, par, lim, 5, 0, mbc, mx, my, q, None, dx, dy, dt)
    h = make(L, 2, (mx, my), 11, 5, 5, par, lim, (dx, dy))
    try:
        L.check(L.lib().pcl_put_q(h, L.d(q), 1))
        cfl = C.c_double()
        L.check(L.lib().pcl_sharp_dq(h, dt, C.cast(C.byref(cfl), L.dp)))
        L.check(L.lib().pcl_select(h, 3))
        out = np.zeros_like(q)
        L.check(L.lib().pcl_get_q(h, L.d(out), 1))
    finally:
        L.lib().pcl_destroy(h)
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    assert np.array_equal(out[inner], ref[inner]), np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("mx", [1, 57, 58, 59, 300])
@pytest.mark.parametrize("lim", [2, 3])
def test_flux1_acoustics1d_bitexact(coracle, mx, lim):
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx)
    mbc = 3
    q = np.asfortranarray(rng.standard_normal((2, mx + 2 * mbc)))
    par = [1.0, 1.0, 1.0, 1.0]
    dx, dt = 1.0 / mx, 0.5 / mx
    ref, cfl_ref = coracle.sharp_flux1(O.RP_ACOUSTICS_1D, par, lim, 2, 0, mbc, mx, q, None, dx, dt)
    h = make(L, 1, (mx,), 2, 2, 2, par, lim, (dx,))
    try:
        L.check(L.lib().pcl_put_q(h, L.d(q), 1))
        cfl = C.c_double()
        L.check(L.lib().pcl_sharp_dq(h, dt, C.cast(C.byref(cfl), L.dp)))
        L.check(L.lib().pcl_select(h, 3))
        out = np.zeros_like(q)
        L.check(L.lib().pcl_get_q(h, L.d(out), 1))
    finally:
        L.lib().pcl_destroy(h)
    assert np.array_equal(out[:, mbc:-mbc], ref[:, mbc:-mbc]), np.abs(out - ref)[:, mbc:-mbc].max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("lim,ti", [(3, 'SSP104'), (2, 'SSP104'), (2, 'SSP33'), (2, 'Euler')])
def test_acoustics2d_sharpclaw_replay(coracle, golden_dir, lim, ti):
    """test/test_examples.py:333-376: SharpClaw 2-D acoustics vs test/ac_sc_solution (gate 1e-4);
    the golden was produced by the legacy weno5 (lim_type=3): 5.6e-14."""
    import pyclaw_amd as pyclaw
    tfinal = 0.12 if ti == 'SSP104' else 0.03
    claw = problems.acoustics2D(pyclaw, solver_type='sharpclaw', lim_type=lim, time_integrator=ti,
                                tfinal=tfinal, nout=10 if ti == 'SSP104' else 2, run=False)
    if ti == 'Euler':
        claw.solver.cfl_max, claw.solver.cfl_desired = 0.2, 0.15
        claw.solver.dt_initial = claw.solver.dt_initial / 3
    claw.run()
    p = D.acoustics2d_problem(solver_type='sharpclaw', lim_type=lim, time_integrator=ti)
    if ti == 'Euler':
        p.cfl_max, p.cfl_desired = 0.2, 0.15
        p.dt_initial = p.dt_initial / 3
    D.run(p, coracle, tfinal, 10 if ti == 'SSP104' else 2)
    q = claw.frames[claw.nout].state.q
    assert np.array_equal(q, p.q), np.abs(q - p.q).max()
    if ti == 'SSP104':
        gold = np.loadtxt(os.path.join(golden_dir, "ac_sc_solution"))
        err = np.linalg.norm(q[0] - gold)
        assert err < 1e-4
        if lim == 3:
            assert err < 1e-13


def test_acoustics1d_sharpclaw_scalar(coracle):
    """test/test_examples.py:103-117: one-period L1 error 0.000298935748775 (gate 1e-5)"""
    import pyclaw_amd as pyclaw
    err, claw = problems.acoustics1D(pyclaw, solver_type='sharpclaw')
    assert abs(err - 0.000298935748775) < 1e-5
    p = D.acoustics1d_problem(solver_type='sharpclaw', cfl_max=2.5, cfl_desired=2.45)
    D.run(p, coracle, 1.0, 5)
    assert np.array_equal(claw.frames[5].state.q, p.q)


def test_sharpclaw_rejected_step(coracle):
    """dt_initial far too large: CFLError inside a stage -> step returns False -> retake"""
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw, mx=40, my=40, solver_type='sharpclaw', tfinal=0.05, nout=1, run=False)
    claw.solver.dt_initial = 0.2
    claw.run()
    p = D.acoustics2d_problem(mx=40, my=40, solver_type='sharpclaw')
    p.dt_initial = 0.2
    D.run(p, coracle, 0.05, 1)
    assert p.nrejected >= 1
    assert np.array_equal(claw.frames[1].state.q, p.q)


def test_sharpclaw_fast_math(coracle):
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw, solver_type='sharpclaw', lim_type=3, math='fast')
    p = D.acoustics2d_problem(solver_type='sharpclaw', lim_type=3)
    D.run(p, coracle, 0.12, 10)
    q = claw.frames[claw.nout].state.q
    assert np.max(np.abs(q - p.q)) < 1e-12 * np.abs(p.q).max()


def test_f2py_shaped_flux2(coracle):
    """layer-1 entry point pcl_sharp_flux2 on host arrays == oracle flux2"""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(77)
    mx, my, mbc = 45, 31, 3
    q = euler_q(rng, (mx + 2 * mbc, my + 2 * mbc))
    par = np.array([1.4, 0.4, 0, 0, 0, 0, 0, 0.])
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.01
    ref, cfl_ref = coracle.sharp_flux2(O.RP_EULER5_2D, par[:2], 2, 5, 0, mbc, mx, my, q, None, dx, dy, dt)
    dq = np.zeros_like(q)
    cfl = C.c_double()
    L.check(L.lib().pcl_sharp_flux2(11, L.d(par), 2, 5, 5, 0, 0, mbc, mx, my, L.d(q), L.d(dq), None, dx, dy, dt,
                                    C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    assert np.array_equal(dq[inner], ref[inner]) and cfl.value == cfl_ref
    assert np.count_nonzero(dq[:, :mbc]) == 0


# ---- char_decomp = 1: wave-based reconstruction in 1-D (1d/sharpclaw/flux1.f90:80-107) ---------------------------------
def state_1d(rp, rng, n):
    if rp == O.RP_EULER_1D:
        rho = 1 + 0.3 * rng.random(n); u = 0.6 * (rng.random(n) - .5); p = 1 + 0.3 * rng.random(n)
        rho[n // 2:] *= 0.4                            # a contact / shock
        return np.asfortranarray(np.stack([rho, rho * u, p / 0.4 + 0.5 * rho * u * u])), [1.4, 0.4]
    if rp == O.RP_SHALLOW_1D:
        h = 1 + 0.3 * rng.random(n); h[n // 3:] += 0.5
        return np.asfortranarray(np.stack([h, h * 0.4 * (rng.random(n) - .5)])), [9.81]
    if rp == O.RP_ACOUSTICS_1D:
        q = rng.standard_normal((2, n)); q[:, n // 4:n // 4 + 9] = 0.25       # a constant stretch: waves of zero norm
        return np.asfortranarray(q), [1.0, 4.0, 2.0, 2.0]
    if rp == O.RP_BURGERS_1D:
        return np.asfortranarray(rng.standard_normal((1, n))), []
    return np.asfortranarray(rng.standard_normal((1, n))), [0.7]              # advection


@pytest.mark.parametrize("rp,meqn,mwaves", [(O.RP_ADVECTION_1D, 1, 1), (O.RP_ACOUSTICS_1D, 2, 2), (O.RP_BURGERS_1D, 1, 1),
                                             (O.RP_EULER_1D, 3, 3), (O.RP_SHALLOW_1D, 2, 2)])
@pytest.mark.parametrize("lim,mth", [(2, 1), (1, 1), (1, 2), (1, 3), (1, 4), (1, 5)])
@pytest.mark.parametrize("mx", [57, 58, 59, 333])
def test_flux1_wave_based_bitexact(coracle, rp, meqn, mwaves, lim, mth, mx):
    """pcl_sharp_flux1 with the module state char_decomp = 1 (pcl_sharp_module_char_decomp) == the oracle's flux1 with
    rp1(q, q) + tvd2_wave (lim_type 1, mthlim per WAVE) / weno5_wave (lim_type 2), bit for bit, Courant number included;
    strip boundaries at 58 cells."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(1000 * rp + mx + lim)
    mbc = 3
    q, par = state_1d(rp, rng, mx + 2 * mbc)
    dx, dt = 1.0 / mx, 0.2 / mx
    coracle.set_char_decomp(1)
    coracle.set_sharp_mthlim([mth] * mwaves)
    try:
        ref, cfl_ref = coracle.sharp_flux1(rp, par + [0.0] * (8 - len(par)), lim, mwaves, 0, mbc, mx, q, None, dx, dt)
    finally:
        coracle.set_char_decomp(0)
        coracle.set_sharp_mthlim([1] * 8)
    dq = np.zeros_like(q)
    cfl = C.c_double()
    L.check(lib.pcl_sharp_module_char_decomp(1))
    L.check(lib.pcl_sharp_module_mthlim(L.i(np.array([mth] * mwaves, dtype=np.int32)), mwaves))
    try:
        L.check(lib.pcl_sharp_flux1(rp, L.d(np.array(par + [0.0] * (8 - len(par)))), lim, meqn, mwaves, 0, 0, mbc, mx, L.d(q),
                                    L.d(dq), None, dx, dt, C.cast(C.byref(cfl), L.dp)))
    finally:
        L.check(lib.pcl_sharp_module_char_decomp(0))
        L.check(lib.pcl_sharp_module_mthlim(L.i(np.ones(8, dtype=np.int32)), 8))
    assert np.isfinite(ref[:, mbc:-mbc]).all() and np.abs(ref[:, mbc:-mbc]).max() > 0
    assert np.array_equal(dq[:, mbc:-mbc], ref[:, mbc:-mbc]), np.abs(dq - ref)[:, mbc:-mbc].max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("lim,ti", [(2, 'SSP104'), (1, 'SSP104')])
def test_acoustics1d_wave_based_replay(coracle, lim, ti):
    """SharpClawSolver1D with char_decomp = 1 through evolve_to_time (fused Runge-Kutta stages) == the oracle driver's
    replay of the same run, bit for bit; the one-period error stays at the component-wise scheme's level"""
    import pyclaw_amd as pyclaw
    err, claw = problems.acoustics1D(pyclaw, solver_type='sharpclaw', lim_type=lim, time_integrator=ti, char_decomp=1)
    kw = dict(cfl_max=2.5, cfl_desired=2.45) if ti == 'SSP104' else dict(cfl_max=2.5, cfl_desired=2.45)
    p = D.acoustics1d_problem(solver_type='sharpclaw', lim_type=lim, time_integrator=ti, char_decomp=1, **kw)
    D.run(p, coracle, 1.0, 5)
    coracle.set_char_decomp(0)
    assert np.array_equal(claw.frames[5].state.q, p.q), np.abs(claw.frames[5].state.q - p.q).max()
    assert err < (5e-2 if lim == 1 else 5e-3), err


def test_char_decomp_refusals():
    import pyclaw_amd as pyclaw
    from pyclaw_amd import _lib as L
    s2 = pyclaw.SharpClawSolver2D()
    s2.char_decomp = 1
    s2.rp = pyclaw.riemann.rp_acoustics_2d
    s2.mwaves = 2
    with pytest.raises(NotImplementedError):
        s2.setup(None)
    for cd in (2, 3):
        s1 = pyclaw.SharpClawSolver1D()
        s1.char_decomp = cd
        with pytest.raises(NotImplementedError):
            s1.setup(None)
    assert L.lib().pcl_sharp_module_char_decomp(2) != 0
