"""
Goldens generated from the REFERENCE'S OWN FORTRAN (flang build, tests/golden/make_ref_goldens.py) for the paths
the reference ships no golden for -- unsplit step2.f (method(3) = 0/1/2, with and without a capacity function),
step2ds.f with a capacity function,
SharpClaw flux2.f90 (lim_type 2 and 3).  CPU: the oracle's C restatement must reproduce them bit for bit;
GPU (-m gpu): the HIP path, through the C ABI, must too.  Unlike tests/test_oracle_vs_ref.py these do not need
oracle/_ref at test time.
"""
import ctypes as C
import importlib.util
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_ref_goldens", os.path.join(HERE, "golden", "make_ref_goldens.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)


def load(name):
    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)


def shapes(z, mbc):
    mx, my = int(z["mx"]), int(z["my"])
    return mx, my, (mx + 2 * mbc, my + 2 * mbc), float(z["dx"]), float(z["dy"]), float(z["dt"])


@pytest.mark.parametrize("trans", [0, 1, 2])
def test_oracle_step2_unsplit(coracle, trans):
    z = load("ref_step2_unsplit.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0 = G.euler_state(10 + trans, shape)
    qn = q0.copy("F")
    method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
    _, cfl = coracle.step2(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method, G.MTH)
    assert np.array_equal(qn, z["q_trans%d" % trans]) and cfl == float(z["cfl_trans%d" % trans])


@pytest.mark.parametrize("ids", [1, 2])
def test_oracle_step2ds_capa(coracle, ids):
    z = load("ref_step2ds_capa.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0, aux = G.euler_state(20 + ids, shape), G.capa_field(20 + ids, shape)
    qn = q0.copy("F")
    method = np.array([1, 2, -1, 0, 0, 2, 2], dtype=np.int32)
    _, cfl = coracle.step2ds(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), qn, aux, dx, dy, dt, method,
                             G.MTH, ids)
    assert np.array_equal(qn, z["q_ids%d" % ids]) and cfl == float(z["cfl_ids%d" % ids])


@pytest.mark.parametrize("lim", [2, 3])
def test_oracle_sharp_flux2(coracle, lim):
    z = load("ref_sharp_flux2.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 3)
    q0 = G.euler_state(30 + lim, shape)
    dq, cfl = coracle.sharp_flux2(O.RP_EULER5_2D, G.PAR, lim, 5, 0, 3, mx, my, q0, None, dx, dy, dt)
    assert np.array_equal(dq[:, 3:-3, 3:-3], z["dq_lim%d" % lim]) and cfl == float(z["cfl_lim%d" % lim])


@pytest.mark.parametrize("trans", [0, 1, 2])
def test_oracle_step2_unsplit_capa(coracle, trans):
    """unsplit step2.f WITH a capacity function (step2.f:145-152,227-234: the annulus / sphere configuration)"""
    z = load("ref_step2_unsplit_capa.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0, aux = G.euler_state(40 + trans, shape), G.capa_field(40 + trans, shape)
    qn = q0.copy("F")
    method = np.array([1, 2, trans, 0, 0, 2, 2], dtype=np.int32)
    _, cfl = coracle.step2(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), qn, aux, dx, dy, dt, method, G.MTH)
    assert np.array_equal(qn, z["q_trans%d" % trans]) and cfl == float(z["cfl_trans%d" % trans])


@pytest.mark.parametrize("which", ["ids1", "ids2", "trans0", "trans1", "trans2"])
def test_oracle_flux2fw_form(coracle, which):
    """flux2fw.f (classic2fw: step2ds.f / step2.f linked with flux2fw.f) built from the reference tree with the vendored
    Euler rpn2/rpt2: the oracle's fwave=1 path must reproduce the reference's own bits.  This pins the ARITHMETIC of
    flux2fw.f:145-152 (dsign(1,s) in the correction); the HIP f-wave kernels are held to the oracle by
    tests/test_fwave.py.  (step1fw.f stays unpinned by a reference build: no rp1 lies in the reference tree.)"""
    z = load("ref_flux2fw.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    if which.startswith("ids"):
        ids = int(which[3:])
        q0 = G.euler_state(70 + ids, shape)
        qn = q0.copy("F")
        method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
        _, cfl = coracle.step2ds(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method,
                                 G.MTH, ids, fwave=True)
        plain = q0.copy("F")
        coracle.step2ds(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), plain, None, dx, dy, dt, method, G.MTH, ids)
    else:
        trans = int(which[5:])
        q0 = G.euler_state(80 + trans, shape)
        qn = q0.copy("F")
        method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
        _, cfl = coracle.step2(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), qn, None, dx, dy, dt, method,
                               G.MTH, fwave=True)
        plain = q0.copy("F")
        coracle.step2(O.RP_EULER5_2D, G.PAR, max(mx, my), 2, mx, my, q0.copy("F"), plain, None, dx, dy, dt, method, G.MTH)
    assert np.array_equal(qn, z["q_" + which]) and cfl == float(z["cfl_" + which])
    assert not np.array_equal(qn, plain)          # the f-wave form really is a different formula


@pytest.mark.parametrize("kind,lim", [(1, 1), (1, 2), (1, 3), (1, 4), (1, 5), (2, 0), (3, 0)])
def test_oracle_wave_based_reconstructions(coracle, kind, lim):
    """char_decomp = 1: tvd2_wave (kind 1, every limiter of its select case incl. Cada-Torrilhon), weno5_wave (2) and
    weno5_fwave (3) of 1d/sharpclaw/reconstruct.f90, called in the reference's own flang build on seeded arrays
    (oracle/ref_sharpclaw_shim.f90: sc_recon_wave; out-of-array reads of the Fortran loops see zero padding): the C
    restatement reproduces ql and qr at EVERY index, bit for bit."""
    z = load("ref_recon_wave.npz")
    q, wave, s = G.recon_wave_inputs(kind, lim)
    ql, qr = coracle.recon_wave(kind, q, wave, s, [max(lim, 1)] * 3)
    assert np.array_equal(ql, z["ql_kind%d_lim%d" % (kind, lim)]) and np.array_equal(qr, z["qr_kind%d_lim%d" % (kind, lim)])
    assert np.isfinite(ql).all() and not np.array_equal(ql[:, 4:-4], q[:, 4:-4])


@pytest.mark.parametrize("mth", [1, 2, 3, 4, 5])
def test_oracle_sharp_tvd2(coracle, mth):
    """lim_type = 1 (tvd2, reconstruct.f90:568-625).  Cells of the first interior row / column are left out: there the
    Fortran reads an uninitialised variable (oracle/sharpclaw_oracle.c: tvd2)."""
    z = load("ref_sharp_tvd2.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 3)
    q0 = G.euler_state(50 + mth, shape)
    coracle.set_sharp_mthlim([mth] * 5)
    dq, _ = coracle.sharp_flux2(O.RP_EULER5_2D, G.PAR, 1, 5, 0, 3, mx, my, q0, None, dx, dy, dt)
    assert np.array_equal(dq[:, 4:-3, 4:-3], z["dq_mth%d" % mth][:, 1:, 1:])


# ------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("mth", [1, 2, 3, 4, 5])
def test_hip_sharp_tvd2(coracle, mth):
    from pyclaw_amd import _lib as L
    z = load("ref_sharp_tvd2.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 3)
    q0 = G.euler_state(50 + mth, shape)
    dq = np.zeros_like(q0)
    cfl = C.c_double()
    mthlim = np.array([mth] * 5, dtype=np.int32)
    L.check(L.lib().pcl_sharp_module_mthlim(L.i(mthlim), 5))
    try:
        L.check(L.lib().pcl_sharp_flux2(O.RP_EULER5_2D, L.d(np.array(G.PAR + [0.0] * 6)), 1, 5, 5, 0, 0, 3, mx, my,
                                        L.d(q0), L.d(dq), None, dx, dy, dt, C.cast(C.byref(cfl), L.dp)))
    finally:
        L.check(L.lib().pcl_sharp_module_mthlim(L.i(np.ones(5, dtype=np.int32)), 5))
    assert np.array_equal(dq[:, 4:-3, 4:-3], z["dq_mth%d" % mth][:, 1:, 1:])     # the reference's own Fortran
    coracle.set_sharp_mthlim([mth] * 5)
    ref, cfl_ref = coracle.sharp_flux2(O.RP_EULER5_2D, G.PAR, 1, 5, 0, 3, mx, my, q0, None, dx, dy, dt)
    assert np.array_equal(dq[:, 3:-3, 3:-3], ref[:, 3:-3, 3:-3]) and cfl.value == cfl_ref   # every cell vs the oracle


@pytest.mark.gpu
@pytest.mark.parametrize("trans", [0, 1, 2])
def test_hip_step2_unsplit_capa(trans):
    from pyclaw_amd import _lib as L
    z = load("ref_step2_unsplit_capa.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0, aux = G.euler_state(40 + trans, shape), G.capa_field(40 + trans, shape)
    out = q0.copy("F")
    method = np.array([1, 2, trans, 0, 0, 2, 2], dtype=np.int32)
    mth = np.array(G.MTH, dtype=np.int32)
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_EULER5_2D, L.d(np.array(G.PAR + [0.0] * 6)), 0, 5, 5, 2, 2, mx, my, L.d(q0), L.d(out), L.d(aux),
                              dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(2, -2), slice(2, -2))
    assert np.array_equal(out[inner], z["q_trans%d" % trans][inner]) and cfl.value == float(z["cfl_trans%d" % trans])


@pytest.mark.gpu
@pytest.mark.parametrize("trans", [0, 1, 2])
def test_hip_step2_unsplit(trans):
    from pyclaw_amd import _lib as L
    z = load("ref_step2_unsplit.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0 = G.euler_state(10 + trans, shape)
    out = q0.copy("F")
    method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array(G.MTH, dtype=np.int32)
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_EULER5_2D, L.d(np.array(G.PAR + [0.0] * 6)), 0, 5, 5, 0, 2, mx, my, L.d(q0), L.d(out), None,
                              dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(2, -2), slice(2, -2))
    assert np.array_equal(out[inner], z["q_trans%d" % trans][inner]) and cfl.value == float(z["cfl_trans%d" % trans])


@pytest.mark.gpu
@pytest.mark.parametrize("ids", [1, 2])
def test_hip_step2ds_capa(ids):
    from pyclaw_amd import _lib as L
    z = load("ref_step2ds_capa.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 2)
    q0, aux = G.euler_state(20 + ids, shape), G.capa_field(20 + ids, shape)
    out = q0.copy("F")
    method = np.array([1, 2, -1, 0, 0, 2, 2], dtype=np.int32)
    mth = np.array(G.MTH, dtype=np.int32)
    cfl = C.c_double()
    L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(np.array(G.PAR + [0.0] * 6)), 0, 5, 5, 2, 2, mx, my, L.d(q0), L.d(out),
                                L.d(aux), dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
    assert np.array_equal(out, z["q_ids%d" % ids]) and cfl.value == float(z["cfl_ids%d" % ids])


@pytest.mark.gpu
@pytest.mark.parametrize("lim", [2, 3])
def test_hip_sharp_flux2(lim):
    from pyclaw_amd import _lib as L
    z = load("ref_sharp_flux2.npz")
    mx, my, shape, dx, dy, dt = shapes(z, 3)
    q0 = G.euler_state(30 + lim, shape)
    dq = np.zeros_like(q0)
    cfl = C.c_double()
    L.check(L.lib().pcl_sharp_flux2(O.RP_EULER5_2D, L.d(np.array(G.PAR + [0.0] * 6)), lim, 5, 5, 0, 0, 3, mx, my,
                                    L.d(q0), L.d(dq), None, dx, dy, dt, C.cast(C.byref(cfl), L.dp)))
    assert np.array_equal(dq[:, 3:-3, 3:-3], z["dq_lim%d" % lim]) and cfl.value == float(z["cfl_lim%d" % lim])
