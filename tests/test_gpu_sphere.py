"""
GPU parity (-m gpu) for shallow water on the sphere (BASELINE configs[4]): rpn2/rpt2_shallow_sphere, the app's
step2qcor.f conservation fix, src2.f and the mirrored y boundary on the device, through the C ABI, against the oracle
bit for bit; the whole reference regression against test/swsphere_height; and the SharpClaw right-hand side with the
real solver on the 2048 x 1024 grid by spot parity.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import driver as D
from oracle import oracle as O

G = 11489.57219


def sphere_fields(coracle, mx, my, mbc, seed, amp=0.05):
    """aux from the oracle's setaux (== the reference's setaux.f bit for bit) and a perturbed Rossby-Haurwitz state
    with its ghost cells filled like the app does (periodic in x, mirrored in y)."""
    dx, dy = 4.0 / mx, 2.0 / my
    aux = coracle.sphere_setaux(mbc, mx, my, -3.0, -1.0, dx, dy)
    q = coracle.sphere_qinit(mbc, mx, my, -3.0, -1.0, dx, dy)
    rng = np.random.default_rng(seed)
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    q[inner] *= 1.0 + amp * (rng.random(q[inner].shape) - 0.5)
    q[:, :mbc, :] = q[:, -2 * mbc:-mbc, :]
    q[:, -mbc:, :] = q[:, mbc:2 * mbc, :]
    for j in range(mbc):
        q[:, :, j] = q[:, ::-1, 2 * mbc - 1 - j]
        q[:, :, my + mbc + j] = q[:, ::-1, my + mbc - 1 - j]
    return np.asfortranarray(q), aux, dx, dy


@pytest.mark.parametrize("mx,my", [(40, 20), (70, 33), (130, 61)])
@pytest.mark.parametrize("trans", [0, 1, 2])
def test_step2qcor_bitexact(coracle, mx, my, trans):
    """pcl_step2 with the sphere solver == step2qcor.f + flux2.f + rpn2/rpt2_shallow_sphere + qcor.f (oracle)"""
    from pyclaw_amd import _lib as L
    q0, aux, dx, dy = sphere_fields(coracle, mx, my, 2, 7 * mx + trans)
    par = np.array([G, dx, dy, 0, 0, 0, 0, 0], dtype=np.float64)
    mth = np.array([4, 4, 4], dtype=np.int32)
    method = np.array([1, 2, trans, 0, 0, 1, 16], dtype=np.int32)
    dt = 0.2 * min(dx, dy) / 4.0
    ref = q0.copy("F")
    coracle.set_qcor(True)
    try:
        _, cfl_ref = coracle.step2(O.RP_SHALLOW_SPHERE_2D, par[:3], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy,
                                   dt, method, mth)
    finally:
        coracle.set_qcor(False)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_SHALLOW_SPHERE_2D, L.d(par), 0, 4, 3, 16, 2, mx, my, L.d(q0), L.d(out), L.d(aux),
                              dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(2, -2), slice(2, -2))
    assert 0.0 < cfl_ref < 1.0
    assert np.array_equal(out[inner], ref[inner]), "max diff %g" % np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("ids", [1, 2])
def test_step2ds_sphere_bitexact(coracle, ids):
    """the dimension-split sweep with the same solver (capacity function, no qcor: step2ds.f has none)"""
    from pyclaw_amd import _lib as L
    mx, my = 70, 33
    q0, aux, dx, dy = sphere_fields(coracle, mx, my, 2, 3 + ids)
    par = np.array([G, dx, dy, 0, 0, 0, 0, 0], dtype=np.float64)
    mth = np.array([4, 4, 4], dtype=np.int32)
    method = np.array([1, 2, -1, 0, 0, 1, 16], dtype=np.int32)
    dt = 0.2 * min(dx, dy) / 4.0
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2ds(O.RP_SHALLOW_SPHERE_2D, par[:3], max(mx, my), 2, mx, my, q0.copy("F"), ref, aux, dx, dy, dt,
                                 method, mth, ids)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2ds(O.RP_SHALLOW_SPHERE_2D, L.d(par), 0, 4, 3, 16, 2, mx, my, L.d(q0), L.d(out), L.d(aux),
                                dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
    assert np.array_equal(out, ref), "max diff %g" % np.abs(out - ref).max()
    assert cfl.value == cfl_ref


def test_coriolis_source_and_mirror_bc_on_device(coracle):
    import pyclaw_amd as pyclaw
    from pyclaw_amd import _lib as L
    from apps import shallow_sphere as S
    mx, my = 40, 20
    q0, aux, dx, dy = sphere_fields(coracle, mx, my, 2, 5)
    claw = S.shallow_sphere(pyclaw, mx, my, run=False, aux_full=aux, q0=q0[:, 2:-2, 2:-2])
    solver, state = claw.solver, claw.solution.state
    solver.setup(claw.solution)
    solver._push(state)
    lib, h = L.lib(), solver._h
    # ghost fill: periodic x, mirrored y (solver.py:354-381 order)
    solver.apply_q_bcs(state)
    got = np.zeros_like(q0)
    L.check(lib.pcl_get_q(h, L.d(got), 1))
    assert np.array_equal(got, q0)
    # src2.f for dt = 0.013
    L.check(lib.pcl_src(h, 2, 0.013, None, 0))
    out = np.zeros((4, mx, my), order="F")
    L.check(lib.pcl_get_q(h, L.d(out), 0))
    ref = np.array(q0[:, 2:-2, 2:-2], order="F")
    coracle.sphere_src2(ref, np.array(aux[:, 2:-2, 2:-2], order="F"), -3.0, -1.0, dx, dy, 0.013)
    assert np.array_equal(out, ref)
    solver.teardown()


def test_swsphere_height_golden_and_oracle_replay(coracle, golden_dir):
    """test/test_examples.py:445-472 (test_2D_shallowwatersphere) through ClawSolver2D on the GPU."""
    import pyclaw_amd as pyclaw
    from apps import shallow_sphere as S
    mx, my = 40, 20
    aux = coracle.sphere_setaux(2, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)
    q0 = coracle.sphere_qinit(2, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)[:, 2:-2, 2:-2]
    claw = S.shallow_sphere(pyclaw, mx, my, aux_full=aux, q0=q0)
    height = claw.frames[claw.nout].state.q[0]
    gold = np.loadtxt(os.path.join(golden_dir, "swsphere_height"))
    assert np.linalg.norm(height - gold) < 1.e-4          # the reference's gate
    p = D.shallow_sphere_problem(coracle)
    st = D.run(p, coracle, 10.0, 10)
    assert claw.solver.status['numsteps'] == st[-1]['numsteps']
    assert np.array_equal(claw.frames[claw.nout].state.q, p.q), np.abs(claw.frames[claw.nout].state.q - p.q).max()
    assert np.linalg.norm(height - gold) < 1.e-14


def test_app_generated_data_passes_the_reference_gate(golden_dir):
    """the same run on the app module's own numpy setaux / qinit (what a user gets)"""
    import pyclaw_amd as pyclaw
    from apps import shallow_sphere as S
    claw = S.shallow_sphere(pyclaw)
    gold = np.loadtxt(os.path.join(golden_dir, "swsphere_height"))
    assert np.linalg.norm(claw.frames[claw.nout].state.q[0] - gold) < 1.e-4


def test_c5_grid_2048x1024_sharpclaw_sphere_spot_parity(coracle):
    """BASELINE configs[4]'s synthetic combination: SharpClaw WENO5 right-hand side (flux2.f90) with the sphere solver,
    its 16 aux components and capacity function on 2048 x 1024; windows of the result == the oracle."""
    from pyclaw_amd import _lib as L
    from apps import shallow_sphere as S
    lib = L.lib()
    nx, ny, mbc = 2048, 1024, 3
    dx, dy = 4.0 / nx, 2.0 / ny
    aux = S.setaux(nx, ny, mbc, -3.0, -1.0, dx, dy)
    q = np.zeros((4, nx + 2 * mbc, ny + 2 * mbc), order="F")
    q[:, mbc:-mbc, mbc:-mbc] = S.qinit(nx, ny, -3.0, -1.0, dx, dy)
    q[:, :mbc, :] = q[:, -2 * mbc:-mbc, :]
    q[:, -mbc:, :] = q[:, mbc:2 * mbc, :]
    for j in range(mbc):
        q[:, :, j] = q[:, ::-1, 2 * mbc - 1 - j]
        q[:, :, ny + mbc + j] = q[:, ::-1, ny + mbc - 1 - j]
    par = np.array([G, dx, dy, 0, 0, 0, 0, 0], dtype=np.float64)
    dt = 0.4 * min(dx, dy) / 4.0
    dq = np.zeros_like(q)
    cflp = C.c_double()
    L.check(lib.pcl_sharp_flux2(O.RP_SHALLOW_SPHERE_2D, L.d(par), 2, 4, 3, 16, 1, mbc, nx, ny, L.d(q), L.d(dq), L.d(aux),
                                dx, dy, dt, C.cast(C.byref(cflp), L.dp)))
    assert np.isfinite(dq[:, 3:-3, 3:-3]).all() and 0 < cflp.value < 2.5 and np.abs(dq[:, 3:-3, 3:-3]).max() > 0
    rng = np.random.default_rng(2)
    w = 24
    wins = [(int(rng.integers(3, nx + 3 - w)), int(rng.integers(3, ny + 3 - w))) for _ in range(6)]
    wins += [(3, 3), (nx + 3 - w, ny + 3 - w), (nx // 2, 3), (3, ny // 2)]
    for (i0, j0) in wins:
        qb = np.array(q[:, i0 - 3:i0 + w + 3, j0 - 3:j0 + w + 3], order="F")
        ab = np.array(aux[:, i0 - 3:i0 + w + 3, j0 - 3:j0 + w + 3], order="F")
        ref, _ = coracle.sharp_flux2(O.RP_SHALLOW_SPHERE_2D, par[:3], 2, 3, 1, 3, w, w, qb, ab, dx, dy, dt)
        assert np.array_equal(dq[:, i0:i0 + w, j0:j0 + w], ref[:, 3:-3, 3:-3]), (i0, j0)
