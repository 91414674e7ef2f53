"""
Unsplit 3-D classic algorithm: step3.f + the transverse half of flux3.f with the restated rpt3 / rptt3 of
the variable-coefficient acoustics equations (third-party, absent from the reference tree).

PINNING.  The reference's golden test/pressure_3D.txt (test_3D_acoustics_heterogeneous: 30^3, order_trans 22, MC,
t = 2; gate 2-norm < 1e-4, test/test_examples.py:497-514) is reproduced by the oracle with 2-norm difference 0.0
-- every digit the file holds -- and by the HIP path bit for bit the same.  flux3.f / step3.f are in the tree but
cannot be built without the three third-party Riemann solvers, so there is no oracle/_ref leg for this row.
"""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import driver as D
from oracle import oracle as O


def test_oracle_replays_pressure_3D(coracle, golden_dir):
    p = D.acoustics3d_problem('het')
    st = D.run(p, coracle, 2.0, 10)
    gold = np.loadtxt(os.path.join(golden_dir, "pressure_3D.txt"))
    pfinal = p.q[0].reshape(-1)
    diff = np.linalg.norm(pfinal - gold)
    assert diff < 1.e-4            # the reference's gate
    assert diff < 1.e-13, diff     # achieved: 0.0
    assert sum(s["numsteps"] for s in st) == 70


def random_3d(rng, n, mbc=2):
    full = tuple(k + 2 * mbc for k in n)
    q = np.asfortranarray(rng.standard_normal((4,) + full))
    aux = np.empty((2,) + full, order="F")
    aux[0] = 1.0 + rng.random(full)          # impedance varies from cell to cell
    aux[1] = 0.5 + rng.random(full)
    return q, aux


@pytest.mark.parametrize("trans", [0, 10, 11, 20, 21, 22])
def test_oracle_unsplit_is_consistent_with_dimsplit_to_first_order(coracle, trans):
    """sanity of the restated step3 on smooth data: one unsplit step and one dimension-split step differ by O(dt^2)"""
    n = (12, 10, 9)
    i, j, k = np.meshgrid(*[np.arange(m + 4) for m in n], indexing="ij")
    q = np.zeros((4,) + tuple(m + 4 for m in n), order="F")
    q[0] = np.sin(0.5 * i) * np.cos(0.4 * j) * np.cos(0.3 * k)
    aux = np.ones((2,) + q.shape[1:], order="F")
    mth = np.array([0, 0], dtype=np.int32)
    d = (0.1, 0.11, 0.12)
    errs = []
    for dt in (0.02, 0.01):
        mu = np.array([1, 2 if trans >= 20 else 1, trans, 0, 0, 0, 2], dtype=np.int32)
        a = q.copy("F")
        coracle.step3(O.RP_VC_ACOUSTICS_3D, 12, 2, n[0], n[1], n[2], q.copy("F"), a, aux, d[0], d[1], d[2], dt, mu, mth)
        md = mu.copy()
        md[2] = -1
        b = q.copy("F")
        for idir, src in ((1, q.copy("F")), (2, b), (3, b)):
            coracle.step3ds(O.RP_VC_ACOUSTICS_3D, 12, 2, n[0], n[1], n[2], src, b, aux, d[0], d[1], d[2], dt, md, mth, idir)
        errs.append(np.abs(a[:, 4:-4, 4:-4, 4:-4] - b[:, 4:-4, 4:-4, 4:-4]).max())
    assert errs[0] < 5e-3 and errs[1] < 0.4 * errs[0]


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("trans,order", [(0, 1), (0, 2), (10, 1), (11, 1), (20, 2), (21, 2), (22, 2)])
@pytest.mark.parametrize("n", [(9, 7, 5), (61, 6, 9), (30, 30, 30), (65, 20, 3)])
def test_hip_step3_bitexact(coracle, trans, order, n):
    """pcl_step3 (classic3.step3) == the oracle's step3.f + flux3.f, every transverse mode of flux3.f:46-73"""
    from pyclaw_amd import _lib as L
    rng = np.random.default_rng(sum(n) + trans)
    q0, aux = random_3d(rng, n)
    d = (1.0 / n[0], 0.9 / n[1], 1.1 / n[2])
    dt = 0.25 * min(d) / 1.5
    method = np.array([1, order, trans, 0, 0, 0, 2], dtype=np.int32)
    mth = np.array([4, 3], dtype=np.int32)
    ref = q0.copy("F")
    _, cfl_ref = coracle.step3(O.RP_VC_ACOUSTICS_3D, max(n), 2, n[0], n[1], n[2], q0.copy("F"), ref, aux, d[0], d[1], d[2], dt,
                               method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step3(O.RP_VC_ACOUSTICS_3D, L.d(np.zeros(8)), 4, 2, 2, 2, n[0], n[1], n[2], L.d(q0), L.d(out), L.d(aux),
                              d[0], d[1], d[2], dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None),) + (slice(2, -2),) * 3
    assert np.array_equal(out[inner], ref[inner]), "max diff %g" % np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


@pytest.mark.gpu
def test_pressure_3D_golden_through_ClawSolver3D(coracle, golden_dir):
    """test/test_examples.py:497-514 (test_3D_acoustics_heterogeneous) on the GPU"""
    import pyclaw_amd as pyclaw
    from apps import problems
    claw = problems.acoustics3D(pyclaw, test='het')
    pfinal = claw.frames[claw.nout].state.q[0, :, :, :].reshape(-1)
    gold = np.loadtxt(os.path.join(golden_dir, "pressure_3D.txt"))
    assert np.linalg.norm(pfinal - gold) < 1.e-4
    p = D.acoustics3d_problem('het')
    D.run(p, coracle, 2.0, 10)
    assert np.array_equal(claw.frames[claw.nout].state.q, p.q)
