"""
GPU parity for the restated third-party Riemann solvers rp1_burgers and rpn2/rpt2_advection (named by the
reference's apps/burgers/1d and apps/advection/2d Makefiles; their source is not in the reference tree and the
reference holds no golden for them: parity is HIP == C restatement on seeded inputs, unpinned at the solver
boundary) -- through every kernel family: classic 1-D, 2-D dimension-split, 2-D unsplit, SharpClaw.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O


@pytest.mark.parametrize("mx", [7, 64, 333])
@pytest.mark.parametrize("lim", [0, 2, 4])
def test_burgers_step1(coracle, mx, lim):
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx + lim)
    q = np.asfortranarray(rng.standard_normal((1, mx + 4)))       # both signs: shocks and transonic rarefactions
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array([lim], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.2 / mx
    ref = q.copy("F")
    _, cfl_ref = coracle.step1(O.RP_BURGERS_1D, [0.0], 2, mx, ref, None, dx, dt, method, mth)
    out = q.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(O.RP_BURGERS_1D, None, 1, 1, 0, 2, mx, L.d(out), None, dx, dt, L.i(method), L.i(mth),
                              C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(out[:, 2:-2], ref[:, 2:-2]) and cfl.value == cfl_ref and cfl.value > 0


@pytest.mark.parametrize("lim_type", [2, 3])
def test_burgers_sharpclaw_flux1(coracle, lim_type):
    from pyclaw_amd import _lib as L
    mx, mbc = 150, 3
    rng = np.random.default_rng(lim_type)
    q = np.asfortranarray(np.sin(np.linspace(0, 7, mx + 6))[None, :] + 0.1 * rng.standard_normal((1, mx + 6)))
    dx, dt = 1.0 / mx, 0.3 / mx
    ref, cfl_ref = coracle.sharp_flux1(O.RP_BURGERS_1D, [0.0], lim_type, 1, 0, mbc, mx, q, None, dx, dt)
    dq = np.zeros_like(q)
    cfl = C.c_double()
    L.check(L.lib().pcl_sharp_flux1(O.RP_BURGERS_1D, None, lim_type, 1, 1, 0, 0, mbc, mx, L.d(q), L.d(dq), None,
                                    dx, dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, mbc:-mbc], ref[:, mbc:-mbc]) and cfl.value == cfl_ref


@pytest.mark.parametrize("uv", [(1.0, 0.5), (-0.7, 1.3), (0.0, -1.0)])
@pytest.mark.parametrize("shape", [(9, 6), (70, 130)])
def test_advection2d_split_and_unsplit(coracle, uv, shape):
    from pyclaw_amd import _lib as L
    mx, my = shape
    rng = np.random.default_rng(mx)
    q0 = np.asfortranarray(rng.random((1, mx + 4, my + 4)))
    par = np.array(list(uv) + [0.0] * 6)
    mth = np.array([4], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.3 / max(mx, my)
    cfl = C.c_double()
    for ids in (1, 2):
        method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_ADVECTION_2D, par, max(mx, my), 2, mx, my, q0.copy("F"), ref, None, dx, dy,
                                     dt, method, mth, ids)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2ds(O.RP_ADVECTION_2D, L.d(par), 0, 1, 1, 0, 2, mx, my, L.d(q0), L.d(out), None, dx,
                                    dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref) and cfl.value == cfl_ref
    for trans in (0, 1, 2):
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2(O.RP_ADVECTION_2D, par, max(mx, my), 2, mx, my, q0.copy("F"), ref, None, dx, dy, dt,
                                   method, mth)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2(O.RP_ADVECTION_2D, L.d(par), 0, 1, 1, 0, 2, mx, my, L.d(q0), L.d(out), None, dx, dy,
                                  dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        inner = (slice(None), slice(2, -2), slice(2, -2))
        assert np.array_equal(out[inner], ref[inner]) and cfl.value == cfl_ref


def test_advection2d_sharpclaw_flux2(coracle):
    from pyclaw_amd import _lib as L
    mx, my, mbc = 60, 45, 3
    rng = np.random.default_rng(8)
    q = np.asfortranarray(rng.random((1, mx + 6, my + 6)))
    par = np.array([0.8, -0.6] + [0.0] * 6)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.01
    ref, cfl_ref = coracle.sharp_flux2(O.RP_ADVECTION_2D, par, 2, 1, 0, mbc, mx, my, q, None, dx, dy, dt)
    dq = np.zeros_like(q)
    cfl = C.c_double()
    L.check(L.lib().pcl_sharp_flux2(O.RP_ADVECTION_2D, L.d(par), 2, 1, 1, 0, 0, mbc, mx, my, L.d(q), L.d(dq), None,
                                    dx, dy, dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, mbc:-mbc, mbc:-mbc], ref[:, mbc:-mbc, mbc:-mbc]) and cfl.value == cfl_ref


def test_burgers_app_shock_speed():
    """1-D Burgers through ClawSolver1D: a right-moving shock between u=1 and u=0 travels at speed 1/2 (Rankine-
    Hugoniot), mass changes only by the boundary fluxes: sanity of the restated solver beyond oracle parity."""
    import pyclaw_amd as pyclaw
    solver = pyclaw.ClawSolver1D()
    solver.rp = pyclaw.riemann.rp_burgers_1d
    solver.mwaves = 1
    solver.limiters = pyclaw.limiters.tvd.MC
    solver.bc_lower[0] = solver.bc_upper[0] = pyclaw.BC.outflow
    x = pyclaw.Dimension('x', 0.0, 1.0, 400)
    state = pyclaw.State(pyclaw.Grid(x), 1)
    xc = state.grid.x.center
    state.q[0, :] = 1.0 * (xc < 0.25)
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal = 0.5
    claw.nout = 1
    solver.dt_initial = 0.001
    claw.run()
    q = claw.frames[1].state.q[0]
    front = xc[np.argmin(np.abs(q - 0.5))]
    assert abs(front - (0.25 + 0.5 * 0.5)) < 2.5 / 400
    assert q.max() <= 1.01 and q.min() >= -0.01                   # no spurious oscillation beyond the MC limiter's


def _euler1d_state(rng, n, kind):
    rho = 0.3 + rng.random(n)
    u = 2.5 * (rng.random(n) - 0.5) if kind == "transonic" else 0.2 * rng.random(n)
    p = 0.2 + rng.random(n)
    if kind == "sod":
        x = np.linspace(0, 1, n)
        rho = np.where(x < 0.5, 3.0, 1.0)
        u = np.where(x < 0.5, 0.9, 0.9)           # with u > 0 the 1-rarefaction is transonic
        p = np.where(x < 0.5, 3.0, 1.0)
    q = np.empty((3, n), order="F")
    q[0], q[1], q[2] = rho, rho * u, p / 0.4 + 0.5 * rho * u * u
    return q


@pytest.mark.parametrize("kind", ["smooth", "transonic", "sod"])
@pytest.mark.parametrize("mx", [50, 257])
def test_euler1d_step1_and_sharpclaw(coracle, kind, mx):
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx)
    par = np.array([1.4, 0.4] + [0.0] * 6)
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array([4, 4, 4], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.15 / mx
    q = _euler1d_state(rng, mx + 4, kind)
    ref = q.copy("F")
    _, cfl_ref = coracle.step1(O.RP_EULER_1D, par, 2, mx, ref, None, dx, dt, method, mth)
    out = q.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(O.RP_EULER_1D, L.d(par), 3, 3, 0, 2, mx, L.d(out), None, dx, dt, L.i(method), L.i(mth),
                              C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(out[:, 2:-2], ref[:, 2:-2]) and cfl.value == cfl_ref and np.isfinite(out).all()
    assert not np.array_equal(out[:, 2:-2], q[:, 2:-2])
    q3 = _euler1d_state(rng, mx + 6, kind)
    ref, cfl_ref = coracle.sharp_flux1(O.RP_EULER_1D, par, 2, 3, 0, 3, mx, q3, None, dx, dt)
    dq = np.zeros_like(q3)
    L.check(L.lib().pcl_sharp_flux1(O.RP_EULER_1D, L.d(par), 2, 3, 3, 0, 0, 3, mx, L.d(q3), L.d(dq), None, dx, dt,
                                    C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3], ref[:, 3:-3]) and cfl.value == cfl_ref


@pytest.mark.parametrize("kind", ["smooth", "transonic"])
def test_shallow1d_step1_and_sharpclaw(coracle, kind):
    from pyclaw_amd import _lib as L
    mx = 300
    rng = np.random.default_rng(4)
    par = np.array([9.81] + [0.0] * 7)
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array([4, 4], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.05 / mx

    def state(n):
        h = 0.5 + rng.random(n)
        u = 8.0 * (rng.random(n) - 0.5) if kind == "transonic" else 0.3 * rng.random(n)
        q = np.empty((2, n), order="F")
        q[0], q[1] = h, h * u
        return q
    q = state(mx + 4)
    ref = q.copy("F")
    _, cfl_ref = coracle.step1(O.RP_SHALLOW_1D, par, 2, mx, ref, None, dx, dt, method, mth)
    out = q.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(O.RP_SHALLOW_1D, L.d(par), 2, 2, 0, 2, mx, L.d(out), None, dx, dt, L.i(method), L.i(mth),
                              C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(out[:, 2:-2], ref[:, 2:-2]) and cfl.value == cfl_ref and np.isfinite(out).all()
    q3 = state(mx + 6)
    ref, cfl_ref = coracle.sharp_flux1(O.RP_SHALLOW_1D, par, 3, 2, 0, 3, mx, q3, None, dx, dt)
    dq = np.zeros_like(q3)
    L.check(L.lib().pcl_sharp_flux1(O.RP_SHALLOW_1D, L.d(par), 3, 2, 2, 0, 0, 3, mx, L.d(q3), L.d(dq), None, dx, dt,
                                    C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3], ref[:, 3:-3]) and cfl.value == cfl_ref


def test_sod_shock_tube_app():
    """Sod's problem through ClawSolver1D + rp_euler_1d: contact and shock positions at t = 0.2 within two cells of
    the exact solution (x_contact = 0.5 + 0.9275 t, x_shock = 0.5 + 1.7522 t), total mass conserved to rounding."""
    import pyclaw_amd as pyclaw
    solver = pyclaw.ClawSolver1D()
    solver.rp = pyclaw.riemann.rp_euler_1d
    solver.mwaves = 3
    solver.limiters = [4, 4, 4]
    solver.bc_lower[0] = solver.bc_upper[0] = pyclaw.BC.outflow
    n = 800
    state = pyclaw.State(pyclaw.Grid(pyclaw.Dimension('x', 0.0, 1.0, n)), 3)
    state.aux_global['gamma'] = 1.4
    state.aux_global['gamma1'] = 0.4
    xc = state.grid.x.center
    rho = np.where(xc < 0.5, 1.0, 0.125)
    p = np.where(xc < 0.5, 1.0, 0.1)
    state.q[0], state.q[1], state.q[2] = rho, 0.0, p / 0.4
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = 0.2, 1
    solver.dt_initial = 1e-4
    claw.run()
    q = claw.frames[1].state.q
    assert abs(q[0].sum() - claw.frames[0].state.q[0].sum()) < 1e-10
    drho = np.abs(np.diff(q[0]))
    right = xc[:-1] > 0.8
    x_shock = xc[:-1][right][np.argmax(drho[right])]
    mid = (xc[:-1] > 0.6) & (xc[:-1] < 0.8)
    x_contact = xc[:-1][mid][np.argmax(drho[mid])]
    assert abs(x_shock - (0.5 + 1.7522 * 0.2)) < 3.0 / n
    assert abs(x_contact - (0.5 + 0.9275 * 0.2)) < 4.0 / n


def _sw_state(rng, shape, kind):
    h = 0.5 + rng.random(shape)
    sc = 8.0 if kind == "transonic" else 0.3
    u = sc * (rng.random(shape) - 0.5)
    v = sc * (rng.random(shape) - 0.5)
    q = np.empty((3,) + shape, order="F")
    q[0], q[1], q[2] = h, h * u, h * v
    return q


@pytest.mark.parametrize("kind", ["smooth", "transonic"])
@pytest.mark.parametrize("shape", [(11, 7), (90, 70)])
def test_shallow2d_all_kernel_families(coracle, kind, shape):
    from pyclaw_amd import _lib as L
    mx, my = shape
    rng = np.random.default_rng(mx + (kind == "transonic"))
    par = np.array([9.81] + [0.0] * 7)
    mth = np.array([4, 4, 4], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.02 / max(mx, my)
    q0 = _sw_state(rng, (mx + 4, my + 4), kind)
    cfl = C.c_double()
    for ids in (1, 2):
        method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_SHALLOW_2D, par, max(mx, my), 2, mx, my, q0.copy("F"), ref, None, dx, dy, dt,
                                     method, mth, ids)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2ds(O.RP_SHALLOW_2D, L.d(par), 0, 3, 3, 0, 2, mx, my, L.d(q0), L.d(out), None, dx, dy,
                                    dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref) and cfl.value == cfl_ref and np.isfinite(out).all()
    for trans in (0, 1, 2):
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2(O.RP_SHALLOW_2D, par, max(mx, my), 2, mx, my, q0.copy("F"), ref, None, dx, dy, dt,
                                   method, mth)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2(O.RP_SHALLOW_2D, L.d(par), 0, 3, 3, 0, 2, mx, my, L.d(q0), L.d(out), None, dx, dy, dt,
                                  L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        inner = (slice(None), slice(2, -2), slice(2, -2))
        assert np.array_equal(out[inner], ref[inner]) and cfl.value == cfl_ref
    q3 = _sw_state(rng, (mx + 6, my + 6), kind)
    ref, cfl_ref = coracle.sharp_flux2(O.RP_SHALLOW_2D, par, 2, 3, 0, 3, mx, my, q3, None, dx, dy, dt)
    dq = np.zeros_like(q3)
    L.check(L.lib().pcl_sharp_flux2(O.RP_SHALLOW_2D, L.d(par), 2, 3, 3, 0, 0, 3, mx, my, L.d(q3), L.d(dq), None, dx, dy,
                                    dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3, 3:-3], ref[:, 3:-3, 3:-3]) and cfl.value == cfl_ref


def test_radial_dam_break_app():
    """apps/shallow/2d style radial dam break through ClawSolver2D (unsplit, transverse corrections): the depth
    stays symmetric under x <-> y to rounding, mass is conserved until the wave reaches the boundary, and the
    bore has moved outwards."""
    import pyclaw_amd as pyclaw
    solver = pyclaw.ClawSolver2D()
    solver.rp = pyclaw.riemann.rp_shallow_2d
    solver.mwaves = 3
    solver.limiters = [4, 4, 4]
    solver.dim_split = False
    solver.order_trans = 2
    for k in range(2):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.outflow
    n = 200
    grid = pyclaw.Grid([pyclaw.Dimension('x', -2.5, 2.5, n), pyclaw.Dimension('y', -2.5, 2.5, n)])
    state = pyclaw.State(grid, 3)
    state.aux_global['g'] = 1.0
    X, Y = grid.c_center
    state.q[0] = 2.0 * (np.sqrt(X ** 2 + Y ** 2) <= 0.5) + 1.0 * (np.sqrt(X ** 2 + Y ** 2) > 0.5)
    state.q[1:] = 0.0
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = 1.0, 1
    solver.dt_initial = 1e-3
    claw.run()
    h0, h = claw.frames[0].state.q[0], claw.frames[1].state.q[0]
    assert np.abs(h - h.T).max() < 1e-11
    assert abs(h.sum() - h0.sum()) < 1e-9 * h0.sum()
    r = np.sqrt(X ** 2 + Y ** 2)
    assert h[r < 0.3].mean() < 1.5 and h[(r > 1.2) & (r < 1.6)].max() > 1.05      # centre dropped, bore outside
    assert np.isfinite(claw.frames[1].state.q).all()


@pytest.mark.parametrize("shape", [(10, 7), (130, 90)])
@pytest.mark.parametrize("lims", [[4, 4], [1, 0]])
def test_vc_acoustics2d_split_and_sharpclaw(coracle, shape, lims):
    """Riemann solver with cell-wise coefficients: the kernels stage the solver's aux components next to q in
    the LDS tile (classic dim-split x/y, SharpClaw with and without a capacity function in a third aux field)."""
    from pyclaw_amd import _lib as L
    mx, my = shape
    rng = np.random.default_rng(mx + lims[0])
    cfl = C.c_double()
    mth = np.array(lims, dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.1 / max(mx, my)
    q0 = np.asfortranarray(rng.standard_normal((3, mx + 4, my + 4)))
    aux = np.asfortranarray(0.5 + 2.0 * rng.random((2, mx + 4, my + 4)))
    method = np.array([1, 2, -1, 0, 0, 0, 2], dtype=np.int32)
    for ids in (1, 2):
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_VC_ACOUSTICS_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy,
                                     dt, method, mth, ids)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2ds(O.RP_VC_ACOUSTICS_2D, None, 0, 3, 2, 2, 2, mx, my, L.d(q0), L.d(out), L.d(aux), dx,
                                    dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref) and cfl.value == cfl_ref and cfl.value > 0
    for mcapa, maux in ((0, 2), (3, 3)):
        q3 = np.asfortranarray(rng.standard_normal((3, mx + 6, my + 6)))
        a3 = np.asfortranarray(0.5 + 2.0 * rng.random((maux, mx + 6, my + 6)))
        ref, cfl_ref = coracle.sharp_flux2(O.RP_VC_ACOUSTICS_2D, [0.0], 2, 2, mcapa, 3, mx, my, q3, a3, dx, dy, dt)
        dq = np.zeros_like(q3)
        L.check(L.lib().pcl_sharp_flux2(O.RP_VC_ACOUSTICS_2D, None, 2, 3, 2, maux, mcapa, 3, mx, my, L.d(q3), L.d(dq),
                                        L.d(a3), dx, dy, dt, C.cast(C.byref(cfl), L.dp)))
        assert np.array_equal(dq[:, 3:-3, 3:-3], ref[:, 3:-3, 3:-3]) and cfl.value == cfl_ref


def test_vc_acoustics2d_uniform_medium_equals_constant_coefficient_golden(golden_dir):
    """With a uniform medium the variable-coefficient solver must do exactly what rpn2_acoustics does: the 2-D
    acoustics regression (test/test_examples.py:239-254) run with rp_vc_acoustics_2d + aux = (zz, cc) equals the
    constant-coefficient run bit for bit -- and so inherits its pin, the reference golden acoustics2D_solution."""
    import os
    import pyclaw_amd as pyclaw
    from apps import problems
    const = problems.acoustics2D(pyclaw)
    claw = problems.acoustics2D(pyclaw, run=False)
    old = claw.solution.state
    state = pyclaw.State(old.grid, 3, 2)
    state.q[...] = old.q
    state.aux[0] = old.aux_global['zz']
    state.aux[1] = old.aux_global['cc']
    claw.solution = pyclaw.Solution(state)
    claw.solver.rp = pyclaw.riemann.rp_vc_acoustics_2d
    for k in range(2):
        claw.solver.aux_bc_lower[k] = claw.solver.aux_bc_upper[k] = pyclaw.BC.outflow
    claw.run()
    assert np.array_equal(claw.frames[claw.nout].state.q, const.frames[const.nout].state.q)
    gold = np.loadtxt(os.path.join(golden_dir, "acoustics2D_solution"))
    assert np.linalg.norm(claw.frames[claw.nout].state.q[0] - gold) < 2e-14


@pytest.mark.parametrize("mx", [40, 301])
def test_advection_color_1d(coracle, mx):
    """rp1_advection_color: velocity per cell edge in aux(1) -- the 1-D kernels with a solver aux plane."""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mx)
    method = np.array([1, 2, 0, 0, 0, 0, 1], dtype=np.int32)
    mth = np.array([3], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.3 / mx
    q = np.asfortranarray(rng.random((1, mx + 4)))
    aux = np.asfortranarray(1.5 * (rng.random((1, mx + 4)) - 0.3))          # both signs
    ref = q.copy("F")
    _, cfl_ref = coracle.step1(O.RP_ADVECTION_COLOR_1D, [0.0], 2, mx, ref, aux, dx, dt, method, mth)
    out = q.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(O.RP_ADVECTION_COLOR_1D, None, 1, 1, 1, 2, mx, L.d(out), L.d(aux), dx, dt, L.i(method),
                              L.i(mth), C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(out[:, 2:-2], ref[:, 2:-2]) and cfl.value == cfl_ref and cfl.value > 0
    q3 = np.asfortranarray(rng.random((1, mx + 6)))
    a3 = np.asfortranarray(1.5 * (rng.random((1, mx + 6)) - 0.3))
    ref, cfl_ref = coracle.sharp_flux1(O.RP_ADVECTION_COLOR_1D, [0.0], 2, 1, 0, 3, mx, q3, a3, dx, dt)
    dq = np.zeros_like(q3)
    L.check(L.lib().pcl_sharp_flux1(O.RP_ADVECTION_COLOR_1D, None, 2, 1, 1, 1, 0, 3, mx, L.d(q3), L.d(dq), L.d(a3), dx,
                                    dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3], ref[:, 3:-3]) and cfl.value == cfl_ref


@pytest.mark.parametrize("shape", [(12, 9), (75, 140)])
@pytest.mark.parametrize("trans", [0, 1, 2])
def test_vc_acoustics2d_unsplit(coracle, shape, trans):
    """unsplit step with rpt2_vc_acoustics: the transverse solver reads the aux values of the slices below and
    above (step2.f:97-101,177-181) -- heterogeneous random medium, device == C restatement"""
    from pyclaw_amd import _lib as L
    mx, my = shape
    rng = np.random.default_rng(mx + trans)
    q0 = np.asfortranarray(rng.standard_normal((3, mx + 4, my + 4)))
    aux = np.asfortranarray(0.5 + 2.0 * rng.random((2, mx + 4, my + 4)))
    method = np.array([1, 2, trans, 0, 0, 0, 2], dtype=np.int32)
    mth = np.array([4, 4], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.1 / max(mx, my)
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2(O.RP_VC_ACOUSTICS_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy, dt,
                               method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_VC_ACOUSTICS_2D, None, 0, 3, 2, 2, 2, mx, my, L.d(q0), L.d(out), L.d(aux), dx, dy, dt,
                              L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(2, -2), slice(2, -2))
    assert np.array_equal(out[inner], ref[inner]) and cfl.value == cfl_ref
    assert not np.array_equal(out[inner], q0[inner])


def test_vc_acoustics2d_unsplit_uniform_medium_equals_constant_coefficient():
    """apps/acoustics/2d style unsplit run (dim_split=0, transverse corrections): with a uniform medium the
    variable-coefficient pair rpn2/rpt2_vc_acoustics equals rpn2/rpt2_acoustics bit for bit."""
    import pyclaw_amd as pyclaw
    from apps import problems
    const = problems.acoustics2D(pyclaw, mx=70, my=60, tfinal=0.1, nout=2, dim_split=0)
    claw = problems.acoustics2D(pyclaw, mx=70, my=60, tfinal=0.1, nout=2, dim_split=0, run=False)
    old = claw.solution.state
    state = pyclaw.State(old.grid, 3, 2)
    state.q[...] = old.q
    state.aux[0] = old.aux_global['zz']
    state.aux[1] = old.aux_global['cc']
    claw.solution = pyclaw.Solution(state)
    claw.solver.rp = pyclaw.riemann.rp_vc_acoustics_2d
    for k in range(2):
        claw.solver.aux_bc_lower[k] = claw.solver.aux_bc_upper[k] = pyclaw.BC.outflow
    claw.run()
    assert claw.solver.status['numsteps'] == const.solver.status['numsteps']
    assert np.array_equal(claw.frames[-1].state.q, const.frames[-1].state.q)


@pytest.mark.parametrize("shape", [(14, 10), (80, 125)])
def test_vc_advection2d_all_kernel_families(coracle, shape):
    """rpn2/rpt2_vc_advection (edge velocities in aux): dim-split with a capacity function in a third aux field
    (the annulus app's set-up), unsplit with transverse terms, SharpClaw -- device == C restatement; and with
    uniform velocities it equals the constant-coefficient solver rpn2/rpt2_advection bit for bit."""
    from pyclaw_amd import _lib as L
    mx, my = shape
    rng = np.random.default_rng(mx)
    cfl = C.c_double()
    mth = np.array([4], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.2 / max(mx, my)
    q0 = np.asfortranarray(rng.random((1, mx + 4, my + 4)))
    aux = np.asfortranarray(np.concatenate([1.5 * (rng.random((2, mx + 4, my + 4)) - 0.4),
                                            0.5 + rng.random((1, mx + 4, my + 4))]))
    method = np.array([1, 2, -1, 0, 0, 3, 3], dtype=np.int32)        # mcapa = 3
    for ids in (1, 2):
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_VC_ADVECTION_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy,
                                     dt, method, mth, ids)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2ds(O.RP_VC_ADVECTION_2D, None, 0, 1, 1, 3, 2, mx, my, L.d(q0), L.d(out), L.d(aux), dx,
                                    dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref) and cfl.value == cfl_ref and cfl.value > 0
    aux2 = np.asfortranarray(aux[:2])
    inner = (slice(None), slice(2, -2), slice(2, -2))
    for trans in (0, 1, 2):
        method = np.array([1, 2, trans, 0, 0, 0, 2], dtype=np.int32)
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2(O.RP_VC_ADVECTION_2D, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux2, dx, dy, dt,
                                   method, mth)
        out = q0.copy("F")
        L.check(L.lib().pcl_step2(O.RP_VC_ADVECTION_2D, None, 0, 1, 1, 2, 2, mx, my, L.d(q0), L.d(out), L.d(aux2), dx, dy,
                                  dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        assert np.array_equal(out[inner], ref[inner]) and cfl.value == cfl_ref
        # uniform velocity field == constant-coefficient advection
        uni = np.empty_like(aux2)
        uni[0], uni[1] = 0.7, -0.4
        a = q0.copy("F")
        L.check(L.lib().pcl_step2(O.RP_VC_ADVECTION_2D, None, 0, 1, 1, 2, 2, mx, my, L.d(q0), L.d(a), L.d(uni), dx, dy,
                                  dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        b = q0.copy("F")
        m0 = method.copy(); m0[6] = 0
        L.check(L.lib().pcl_step2(O.RP_ADVECTION_2D, L.d(np.array([0.7, -0.4] + [0.0] * 6)), 0, 1, 1, 0, 2, mx, my,
                                  L.d(q0), L.d(b), None, dx, dy, dt, L.i(m0), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        assert np.array_equal(a[inner], b[inner])
    q3 = np.asfortranarray(rng.random((1, mx + 6, my + 6)))
    a3 = np.asfortranarray(1.5 * (rng.random((2, mx + 6, my + 6)) - 0.4))
    ref, cfl_ref = coracle.sharp_flux2(O.RP_VC_ADVECTION_2D, [0.0], 2, 1, 0, 3, mx, my, q3, a3, dx, dy, dt)
    dq = np.zeros_like(q3)
    L.check(L.lib().pcl_sharp_flux2(O.RP_VC_ADVECTION_2D, None, 2, 1, 1, 2, 0, 3, mx, my, L.d(q3), L.d(dq), L.d(a3), dx,
                                    dy, dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3, 3:-3], ref[:, 3:-3, 3:-3]) and cfl.value == cfl_ref


@pytest.mark.parametrize("rp_name", ["vc_advection", "vc_acoustics"])
@pytest.mark.parametrize("trans", [0, 2])
def test_vc_solvers_unsplit_with_capacity_function(coracle, rp_name, trans):
    """the annulus app's configuration (apps/advection/2d/annulus: unsplit, order_trans=2, mcapa=2 -> third aux
    component): the LDS-exchange unsplit kernels with CAPA, every increment divided by capa of its target cell"""
    from pyclaw_amd import _lib as L
    mx, my = 44, 57
    rng = np.random.default_rng(trans)
    meqn, mw, rp = (1, 1, O.RP_VC_ADVECTION_2D) if rp_name == "vc_advection" else (3, 2, O.RP_VC_ACOUSTICS_2D)
    q0 = np.asfortranarray(rng.standard_normal((meqn, mx + 4, my + 4)))
    first = 1.5 * (rng.random((2, mx + 4, my + 4)) - 0.4) if meqn == 1 else 0.5 + 2.0 * rng.random((2, mx + 4, my + 4))
    aux = np.asfortranarray(np.concatenate([first, 0.5 + rng.random((1, mx + 4, my + 4))]))
    method = np.array([1, 2, trans, 0, 0, 3, 3], dtype=np.int32)
    mth = np.array([4] * mw, dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.002
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2(rp, [0.0], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy, dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(rp, None, 0, meqn, mw, 3, 2, mx, my, L.d(q0), L.d(out), L.d(aux), dx, dy, dt,
                              L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(2, -2), slice(2, -2))
    assert np.array_equal(out[inner], ref[inner]) and cfl.value == cfl_ref and cfl.value > 0


@pytest.mark.parametrize("solver_type", ["classic", "sharpclaw"])
def test_rotating_flow_app_with_capa(coracle, solver_type):
    """Solid-body rotation of a blob, colour equation with edge velocities from a stream function and a capacity
    function in aux(3) (the ingredients of apps/advection/2d/annulus, on a Cartesian grid): the solver classes'
    aux / capa plumbing (aux BCs at setup, upload, mcapa, unsplit + transverse or SharpClaw) against the oracle
    driver's replay, bit for bit, plus (classic) conservation of the capacity-weighted mass."""
    import pyclaw_amd as pyclaw
    from oracle import driver as D
    n = 64
    if solver_type == "classic":
        solver = pyclaw.ClawSolver2D()
        solver.dim_split = False
        solver.order_trans = 2
        solver.limiters = pyclaw.limiters.tvd.vanleer
    else:
        solver = pyclaw.SharpClawSolver2D()
        solver.lim_type = 2
    solver.rp = pyclaw.riemann.rp_vc_advection_2d
    solver.mwaves = 1
    for k in range(2):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.periodic
        solver.aux_bc_lower[k] = solver.aux_bc_upper[k] = pyclaw.BC.periodic
    grid = pyclaw.Grid([pyclaw.Dimension('x', -1.0, 1.0, n), pyclaw.Dimension('y', -1.0, 1.0, n)])
    state = pyclaw.State(grid, 1, 3)
    state.mcapa = 2
    d = grid.d[0]
    xe, ye = grid.x.edge, grid.y.edge
    psi = lambda x, y: 0.5 * np.pi * (np.cos(np.pi * x / 2) ** 2) * (np.cos(np.pi * y / 2) ** 2)   # stream function
    XE, YE = np.meshgrid(xe, ye, indexing="ij")
    P = psi(XE, YE)
    state.aux[0] = (P[:-1, 1:] - P[:-1, :-1]) / d          # u at the left edge   =  d(psi)/dy
    state.aux[1] = -(P[1:, :-1] - P[:-1, :-1]) / d         # v at the bottom edge = -d(psi)/dx
    X, Y = grid.c_center
    state.aux[2] = 1.0 + 0.2 * np.sin(np.pi * X) * np.sin(np.pi * Y)    # capacity
    state.q[0] = np.exp(-40.0 * ((X - 0.3) ** 2 + Y ** 2))
    q0, a0 = state.q.copy(), state.aux.copy()
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = 0.3, 1
    solver.dt_initial = 0.005
    if solver_type == "classic":
        solver.cfl_max, solver.cfl_desired = 1.0, 0.9
    claw.run()
    q = claw.frames[1].state.q
    mass0, mass1 = (q0[0] * a0[2]).sum(), (q[0] * a0[2]).sum()
    if solver_type == "classic":        # wave propagation with edge velocities from a stream function is conservative;
        assert abs(mass1 - mass0) < 1e-11 * abs(mass0)   # SharpClaw's in-cell fluctuation of the colour form is not
    assert np.abs(q - q0).max() > 0.05                                    # it moved
    kw = dict(solver_type=solver_type, lim_type=2) if solver_type == "sharpclaw" else {}
    p = D.Problem(q=np.asfortranarray(q0), aux=np.asfortranarray(a0), rp=O.RP_VC_ADVECTION_2D, rp_params=np.zeros(8),
                  mwaves=1, limiters=3, bc_lower=[D.PERIODIC] * 2, bc_upper=[D.PERIODIC] * 2,
                  aux_bc_lower=[D.PERIODIC] * 2, aux_bc_upper=[D.PERIODIC] * 2, d=(d, d), dim_split=False,
                  order_trans=2, mcapa=2, dt_initial=0.005,
                  **(dict(cfl_max=2.5, cfl_desired=2.45, **kw) if solver_type == "sharpclaw" else
                     dict(cfl_max=1.0, cfl_desired=0.9)))
    st = D.run(p, coracle, 0.3, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps']
    assert np.array_equal(q, p.q)
