"""
GPU: classic solvers with more than the default two ghost layers (`solver.mbc = 3, 4, 5`; the reference lets the user
set it, clawpack.py:94-108).  The strips keep their 2-cell halo; the sweeps cover / copy through the wider frame
(step2ds.f sweeps every ghost row).  HIP == oracle bit for bit through the f2py-shaped ABI, and an app replay.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import driver as D
from oracle import oracle as O


def euler(rng, shape):
    q = np.empty((5,) + shape, order="F")
    rho = 0.5 + rng.random(shape)
    u = rng.random(shape) - .5
    v = rng.random(shape) - .5
    p = 0.5 + rng.random(shape)
    q[0], q[1], q[2] = rho, rho * u, rho * v
    q[3] = p / .4 + .5 * rho * (u * u + v * v)
    q[4] = rng.random(shape)
    return q


@pytest.mark.parametrize("mbc", [3, 4, 5])
@pytest.mark.parametrize("mx,my", [(61, 40), (130, 17), (7, 5), (242, 64)])
def test_step2ds_and_step2_with_more_ghost_layers(coracle, mbc, mx, my):
    from pyclaw_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(mbc * 100 + mx)
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler(rng, shape)
    par = np.array([1.4, .4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    dx, dy, dt = 1. / mx, .7 / my, .05 / max(mx, my)
    for ids in (1, 2):
        method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cr = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None, dx, dy, dt, method, mth, ids)
        out = q0.copy("F")
        cfl = C.c_double()
        L.check(lib.pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out), None, dx, dy, dt,
                                L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref) and cfl.value == cr, (ids, np.nanmax(np.abs(out - ref)))
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    for trans in (0, 1, 2):
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        ref = q0.copy("F")
        _, cr = coracle.step2(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None, dx, dy, dt, method, mth)
        out = q0.copy("F")
        cfl = C.c_double()
        L.check(lib.pcl_step2(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out), None, dx, dy, dt,
                              L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
        assert np.array_equal(out[inner], ref[inner]) and cfl.value == cr, trans


@pytest.mark.parametrize("mbc", [3, 4])
@pytest.mark.parametrize("mx", [1, 59, 60, 61, 300])
def test_step1_with_more_ghost_layers(coracle, mbc, mx):
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(mbc + mx)
    q0 = np.asfortranarray(rng.standard_normal((2, mx + 2 * mbc)))
    par = np.array([1.0, 1.0, 1.0, 1.0])
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array([4, 3], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.4 / mx
    ref = q0.copy("F")
    _, cr = coracle.step1(O.RP_ACOUSTICS_1D, par, mbc, mx, ref, None, dx, dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(O.RP_ACOUSTICS_1D, L.d(par), 2, 2, 0, mbc, mx, L.d(out), None, dx, dt, L.i(method), L.i(mth),
                              C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(out[:, mbc:-mbc], ref[:, mbc:-mbc]) and cfl.value == cr


@pytest.mark.parametrize("dim_split", [True, False])
def test_shockbubble_app_with_three_ghost_layers(coracle, dim_split):
    """the reference regression's set-up with solver.mbc = 3: product == oracle replay bit for bit, and (the scheme
    reaches two cells) the interior result equals the mbc = 2 run"""
    import pyclaw_amd as pyclaw
    from apps import problems
    res = {}
    for mbc in (2, 3):
        claw = problems.shockbubble(pyclaw, tfinal=0.03, device_callbacks=True, dim_split=dim_split, run=False)
        claw.solver.mbc = mbc
        claw.run()
        res[mbc] = claw.frames[claw.nout].state.q.copy()
    p = D.shockbubble_problem(dim_split=dim_split, mbc=3)
    D.run(p, coracle, 0.03, 1)
    assert np.array_equal(res[3], p.q)
    assert np.array_equal(res[3], res[2])
