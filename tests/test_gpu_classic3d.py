"""
GPU parity, 3-D dimension-split classic sweeps: pcl_step3ds (C ABI, f2py-shaped) against
the C restatement of step3ds.f/flux3.f on the same seeded inputs, bit for bit.  The Riemann solver
(rpn3_vc_acoustics, third-party, absent from the reference tree) is restated on both sides: parity is
pinned at the app level only, through the reference's scalar result (tests/test_oracle_golden.py).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O


def vc_state(rng, shape, mbc=2):
    full = tuple(n + 2 * mbc for n in shape)
    q = np.asfortranarray(rng.standard_normal((4,) + full))
    aux = np.empty((2,) + full, order="F")
    aux[0] = 0.5 + 2.0 * rng.random(full)      # impedance
    aux[1] = 0.5 + 1.5 * rng.random(full)      # sound speed
    return q, aux


@pytest.mark.parametrize("shape", [(9, 7, 5), (70, 20, 3), (5, 66, 4), (6, 3, 130), (1, 1, 1)])
@pytest.mark.parametrize("idir", [1, 2, 3])
@pytest.mark.parametrize("order,lims", [(2, [4, 4]), (2, [1, 3]), (2, [2, 0]), (1, [4, 4])])
def test_step3ds_equals_oracle(coracle, shape, idir, order, lims):
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(100 * idir + shape[0])
    mx, my, mz = shape
    q, aux = vc_state(rng, shape)
    method = np.array([1, order, -1, 0, 0, 0, 2], dtype=np.int32)
    mthlim = np.array(lims, dtype=np.int32)
    d = (0.1, 0.07, 0.13)
    dt = 0.02
    want = q.copy("F")
    _, cfl_o = coracle.step3ds(O.RP_VC_ACOUSTICS_3D, max(shape), 2, mx, my, mz, q.copy("F"), want, aux,
                               d[0], d[1], d[2], dt, method, mthlim, idir)
    got = np.zeros_like(q)
    cfl = C.c_double()
    L.check(L.lib().pcl_step3ds(O.RP_VC_ACOUSTICS_3D, None, 4, 2, 2, 2, mx, my, mz, L.d(q), L.d(got), L.d(aux),
                                d[0], d[1], d[2], dt, L.i(method), L.i(mthlim), C.cast(C.byref(cfl), L.dp), idir))
    assert cfl.value == cfl_o and cfl.value > 0
    assert np.array_equal(got, want)
    assert not np.array_equal(got, q)
    # untouched: ghost cells along the sweep and the outer transverse ghost layer
    idx = [slice(None)] * 4
    idx[idir] = slice(0, 2)
    assert np.array_equal(got[tuple(idx)], q[tuple(idx)])


def test_acoustics3d_hom_app(coracle):
    """test/test_examples.py:481-488 (3-D acoustics, homogeneous, dim-split 256x4x4, periodic): the reference
    gates final_difference = 0.00286 +- 1e-4.  The product run must also equal the oracle driver's replay of
    the same script bit for bit (same accept/reject sequence, same dt history)."""
    import pyclaw_amd as pyclaw
    from apps import problems
    from oracle import driver as D
    claw = problems.acoustics3D(pyclaw)
    pinitial = claw.frames[0].state.q[0].reshape(-1)
    pfinal = claw.frames[claw.nout].state.q[0].reshape(-1)
    final_difference = np.prod(claw.solution.state.grid.d) * np.linalg.norm(pfinal - pinitial, ord=1)
    assert abs(final_difference - 0.00286) < 1e-4
    p = D.acoustics3d_problem('hom')
    st = D.run(p, coracle, 2.0, 10)
    assert claw.solver.status['numsteps'] == st[-1]['numsteps']
    assert claw.solver.status['cflmax'] == st[-1]['cflmax']
    assert np.array_equal(claw.frames[claw.nout].state.q, p.q)


def test_solver3d_reflecting_outflow_bcs(coracle):
    """device ghost fills in all three dimensions (reflecting lower / outflow upper) against the oracle driver"""
    import pyclaw_amd as pyclaw
    from apps import problems
    from oracle import driver as D
    claw = problems.acoustics3D(pyclaw, mx=20, my=12, mz=9, run=False, tfinal=0.3, nout=2)
    p = D.acoustics3d_problem('hom', mx=20, my=12, mz=9)
    rng = np.random.default_rng(5)
    q0 = rng.standard_normal(p.q.shape)
    a0 = np.stack([1 + rng.random(p.q.shape[1:]), 0.5 + rng.random(p.q.shape[1:])])
    claw.solution.state.q[...] = q0
    claw.solution.state.aux[...] = a0
    p.q[...] = q0
    p.aux[...] = a0
    for k in range(3):
        claw.solver.bc_lower[k] = claw.solver.aux_bc_lower[k] = pyclaw.BC.reflecting
        claw.solver.bc_upper[k] = claw.solver.aux_bc_upper[k] = pyclaw.BC.outflow
    p.bc_lower = p.aux_bc_lower = [D.REFLECTING] * 3
    p.bc_upper = p.aux_bc_upper = [D.OUTFLOW] * 3
    claw.solver.dt_initial = p.dt_initial = 0.01
    claw.run()
    D.run(p, coracle, 0.3, 2)
    assert np.array_equal(claw.frames[-1].state.q, p.q)
