"""
GPU parity tests (run with -m gpu): the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs -- bit for bit (PCL_MATH_EXACT) -- and against the
reference's own golden files.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O


def _lib():
    from pyclaw_amd import _lib as L
    return L


def euler_state(rng, shape):
    """Smooth random Euler state of SURVEY 8d (rho=1+.1U, ...)."""
    q = np.empty((5,) + shape, order="F")
    q[0] = 1.0 + 0.1 * rng.random(shape)
    q[1] = 0.1 * rng.random(shape) - 0.03
    q[2] = 0.05 * rng.random(shape) - 0.02
    q[3] = 2.5 + 0.1 * rng.random(shape)
    q[4] = rng.random(shape)
    return q


def euler_transonic_state(rng, shape):
    """Strong jumps + supersonic patches: exercises every entropy-fix branch."""
    q = euler_state(rng, shape)
    rho = 0.2 + 2.0 * rng.random(shape)
    u = 3.0 * (rng.random(shape) - 0.5)
    v = 3.0 * (rng.random(shape) - 0.5)
    p = 0.1 + 2.0 * rng.random(shape)
    q[0] = rho
    q[1] = rho * u
    q[2] = rho * v
    q[3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    return q


METHOD_DS = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)


def test_wave_shift_semantics():
    L = _lib()
    x = np.arange(64, dtype=np.float64) * 1.5 + 0.25
    l = np.zeros(64)
    r = np.zeros(64)
    L.check(L.lib().pcl_debug_wave_shift(L.d(x), L.d(l), L.d(r)))
    assert np.array_equal(l[1:], x[:-1]) and l[0] == 0.0      # bound_ctrl: no source lane -> 0
    assert np.array_equal(r[:-1], x[1:]) and r[63] == 0.0


@pytest.mark.parametrize("mx,my", [(1, 1), (7, 5), (60, 60), (61, 59), (130, 75), (257, 300)])
@pytest.mark.parametrize("ids", [1, 2])
@pytest.mark.parametrize("kind", ["smooth", "transonic"])
def test_step2ds_euler_bitexact(coracle, mx, my, ids, kind):
    """pcl_step2ds (f2py-shaped ABI) == oracle step2ds, Euler 5-wave, MC/superbee limiters."""
    L = _lib()
    rng = np.random.default_rng(100 * mx + my + ids)
    mbc = 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler_state(rng, shape) if kind == "smooth" else euler_transonic_state(rng, shape)
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 0.7 / my, 0.1 / max(mx, my)
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None,
                                 dx, dy, dt, METHOD_DS, mth, ids)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out),
                                None, dx, dy, dt, L.i(METHOD_DS), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
    assert np.array_equal(out, ref), "max diff %g" % np.abs(out - ref).max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("mx,my", [(9, 4), (64, 33), (200, 121)])
@pytest.mark.parametrize("lim", [[4, 4], [1, 2], [3, 5], [0, 0]])
@pytest.mark.parametrize("order", [1, 2])
def test_step2ds_acoustics_bitexact(coracle, mx, my, lim, order):
    L = _lib()
    rng = np.random.default_rng(mx * 7 + my)
    mbc = 2
    q0 = np.asfortranarray(rng.standard_normal((3, mx + 2 * mbc, my + 2 * mbc)))
    par = np.array([1.0, 4.0, 2.0, 2.0])
    mth = np.array(lim, dtype=np.int32)
    method = METHOD_DS.copy()
    method[1] = order
    dx, dy, dt = 2.0 / mx, 2.0 / my, 0.2 / max(mx, my)
    for ids in (1, 2):
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_ACOUSTICS_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref,
                                     None, dx, dy, dt, method, mth, ids)
        out = q0.copy("F")
        cfl = C.c_double()
        L.check(L.lib().pcl_step2ds(O.RP_ACOUSTICS_2D, L.d(par), 0, 3, 2, 0, mbc, mx, my, L.d(q0),
                                    L.d(out), None, dx, dy, dt, L.i(method), L.i(mth),
                                    C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref), "ids %d max diff %g" % (ids, np.abs(out - ref).max())
        assert cfl.value == cfl_ref


@pytest.mark.parametrize("mx,my", [(33, 20), (128, 70)])
def test_step2ds_capa_bitexact(coracle, mx, my):
    """capacity function path (mcapa>0): dtdx1d = dtdx/aux(mcapa), update divided by capa."""
    L = _lib()
    rng = np.random.default_rng(5)
    mbc = 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler_state(rng, shape)
    aux = np.asfortranarray(0.5 + rng.random((2,) + shape))
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    method = METHOD_DS.copy()
    method[5] = 2
    method[6] = 2
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.05 / max(mx, my)
    for ids in (1, 2):
        ref = q0.copy("F")
        _, cfl_ref = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, aux,
                                     dx, dy, dt, method, mth, ids)
        out = q0.copy("F")
        cfl = C.c_double()
        L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 2, mbc, mx, my, L.d(q0), L.d(out),
                                    L.d(aux), dx, dy, dt, L.i(method), L.i(mth),
                                    C.cast(C.byref(cfl), L.dp), ids))
        assert np.array_equal(out, ref), "ids %d max diff %g" % (ids, np.abs(out - ref).max())
        assert cfl.value == cfl_ref


@pytest.mark.parametrize("rp,meqn,mwaves,par", [(O.RP_ADVECTION_1D, 1, 1, [1.0]),
                                               (O.RP_ADVECTION_1D, 1, 1, [-0.7]),
                                               (O.RP_ACOUSTICS_1D, 2, 2, [1.0, 1.0, 1.0, 1.0])])
@pytest.mark.parametrize("mx", [1, 59, 60, 61, 1000])
@pytest.mark.parametrize("lim", [1, 2, 3, 4, 5, 0])
def test_step1_bitexact(coracle, rp, meqn, mwaves, par, mx, lim):
    L = _lib()
    rng = np.random.default_rng(mx + lim)
    mbc = 2
    q0 = np.asfortranarray(rng.standard_normal((meqn, mx + 2 * mbc)))
    par = np.array(par + [0.0] * (8 - len(par)))
    mth = np.array([lim] * mwaves, dtype=np.int32)
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    dx, dt = 1.0 / mx, 0.8 / mx
    ref = q0.copy("F")
    _, cfl_ref = coracle.step1(rp, par, mbc, mx, ref, None, dx, dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step1(rp, L.d(par), meqn, mwaves, 0, mbc, mx, L.d(out), None, dx, dt,
                              L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    # interior cells only: the Fortran also dirties ghost cells 0 and mx+1, which nobody reads
    assert np.array_equal(out[:, mbc:-mbc], ref[:, mbc:-mbc]), np.abs(out - ref)[:, mbc:-mbc].max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("mx,my", [(1, 1), (6, 4), (61, 59), (130, 75), (250, 97)])
@pytest.mark.parametrize("trans", [0, 1, 2])
@pytest.mark.parametrize("kind", ["smooth", "transonic"])
def test_step2_unsplit_euler_bitexact(coracle, mx, my, trans, kind):
    """pcl_step2 (unsplit, transverse Riemann solves) == oracle step2 (== reference step2.f, see
    tests/test_oracle_vs_ref.py), interior cells, bit for bit."""
    L = _lib()
    rng = np.random.default_rng(31 * mx + my + trans)
    mbc = 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler_state(rng, shape) if kind == "smooth" else euler_transonic_state(rng, shape)
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    method = np.array([1, 2, trans, 0, 0, 0, 0], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 0.7 / my, 0.05 / max(mx, my)
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None, dx, dy,
                               dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out), None, dx,
                              dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    assert np.array_equal(out[inner], ref[inner]), "max diff %g" % np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("trans", [1, 2])
@pytest.mark.parametrize("order", [1, 2])
def test_step2_unsplit_acoustics_bitexact(coracle, trans, order):
    """restated rpt2_acoustics (no reference golden: parity unpinned at the solver boundary)"""
    L = _lib()
    rng = np.random.default_rng(9)
    mx, my, mbc = 70, 41, 2
    q0 = np.asfortranarray(rng.standard_normal((3, mx + 2 * mbc, my + 2 * mbc)))
    par = np.array([1.0, 4.0, 2.0, 2.0])
    mth = np.array([4, 4], dtype=np.int32)
    method = np.array([1, order, trans, 0, 0, 0, 0], dtype=np.int32)
    dx, dy, dt = 2.0 / mx, 2.0 / my, 0.2 / max(mx, my)
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2(O.RP_ACOUSTICS_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None, dx,
                               dy, dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_ACOUSTICS_2D, L.d(par), 0, 3, 2, 0, mbc, mx, my, L.d(q0), L.d(out), None,
                              dx, dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    assert np.array_equal(out[inner], ref[inner]), np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


def test_step2_unsplit_capa_bitexact(coracle):
    L = _lib()
    rng = np.random.default_rng(12)
    mx, my, mbc = 64, 30, 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler_state(rng, shape)
    aux = np.asfortranarray(0.5 + rng.random((2,) + shape))
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    method = np.array([1, 2, 2, 0, 0, 1, 2], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 1.0 / my, 0.03 / mx
    ref = q0.copy("F")
    _, cfl_ref = coracle.step2(O.RP_EULER5_2D, par, mx, mbc, mx, my, q0.copy("F"), ref, aux, dx, dy, dt, method, mth)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 2, mbc, mx, my, L.d(q0), L.d(out), L.d(aux), dx,
                              dy, dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp)))
    inner = (slice(None), slice(mbc, -mbc), slice(mbc, -mbc))
    assert np.array_equal(out[inner], ref[inner]), np.abs(out[inner] - ref[inner]).max()
    assert cfl.value == cfl_ref


@pytest.mark.parametrize("ids", [1, 2])
@pytest.mark.parametrize("case", ["patches", "uniform", "unphysical"])
def test_jump_free_wavefronts_bitexact(coracle, ids, case):
    """Wavefronts whose cells are all equal take a shortcut (only the wave speeds are computed, the update is
    the identity): piecewise-constant states -- patches of constant gas next to random ones, a fully uniform
    state (the Courant number must still come out identical), and a constant UNPHYSICAL state (negative
    pressure => NaN speeds => the shortcut must stand back: the patch comes out NaN like the reference's.
    NaN-for-NaN equality with the oracle is not asserted there: max/min of a NaN differ between v_max_f64
    and Fortran's dmax1)."""
    L = _lib()
    rng = np.random.default_rng(17 + ids)
    mx, my, mbc = 300, 140, 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    q0 = euler_transonic_state(rng, shape)
    const = np.array([1.3, 0.4, -0.2, 2.9, 0.5])
    if case == "patches":
        q0[:, 70:250, 20:110] = const[:, None, None]              # a block of constant gas, moving
        q0[:, :40, :] = np.array([0.1, 0.0, 0.0, 2.5, 1.0])[:, None, None]   # a still strip incl. ghost cells
    elif case == "uniform":
        q0[...] = const[:, None, None]
    else:
        q0[:, 100:260, 30:100] = np.array([1.0, 0.5, 0.5, 0.1, 0.0])[:, None, None]   # p < 0
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 0.7 / my, 0.02 / max(mx, my)
    ref = q0.copy("F")
    with np.errstate(all="ignore"):
        _, cfl_ref = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None,
                                     dx, dy, dt, METHOD_DS, mth, ids)
    out = q0.copy("F")
    cfl = C.c_double()
    L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out),
                                None, dx, dy, dt, L.i(METHOD_DS), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
    if case == "unphysical":
        inner = (slice(0, 4), slice(110, 250), slice(40, 90))    # (the tracer's known-zero wave entries are
        assert np.isnan(ref[inner]).all() and np.isnan(out[inner]).all()   # skipped on the device: 0, not NaN*0)
        ok = ~(np.isnan(ref) | np.isnan(out))
        assert np.array_equal(out[ok], ref[ok])
    else:
        assert np.array_equal(out, ref), "max diff %g" % np.abs(out - ref).max()
        assert cfl.value == cfl_ref and cfl.value > 0
    if case == "uniform":
        assert np.array_equal(out, q0)


def test_layer1_aliased_qnew_and_cached_handle(coracle):
    """The reference's second step2ds call passes ONE array as qold and qnew (clawpack.py:542-543); the f2py-shaped entry
    must take that, keep its device buffers between calls of the same shape, and refuse a qnew that differs from qold
    on entry (the Fortran would add to it; no caller of the reference does that)."""
    L = _lib()
    rng = np.random.default_rng(8)
    mx, my, mbc = 70, 41, 2
    shape = (mx + 2 * mbc, my + 2 * mbc)
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 0.7 / my, 0.1 / max(mx, my)
    q = euler_state(rng, shape)
    ref = q.copy("F")
    cfl = C.c_double()
    for step in range(3):                                   # x sweep (copy), y sweep (aliased), three times
        qold = ref.copy("F")
        _, c1 = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, qold, ref, None, dx, dy, dt, METHOD_DS, mth, 1)
        _, c2 = coracle.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, ref, ref, None, dx, dy, dt, METHOD_DS, mth, 2)
        qold = q.copy("F")
        L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(qold), L.d(q), None, dx, dy, dt,
                                    L.i(METHOD_DS), L.i(mth), C.cast(C.byref(cfl), L.dp), 1))
        assert cfl.value == c1
        L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q), L.d(q), None, dx, dy, dt,
                                    L.i(METHOD_DS), L.i(mth), C.cast(C.byref(cfl), L.dp), 2))
        assert cfl.value == c2
        assert np.array_equal(q, ref), step
    other = q + 1.0
    rc = L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q), L.d(other), None, dx, dy, dt,
                             L.i(METHOD_DS), L.i(mth), C.cast(C.byref(cfl), L.dp), 1)
    assert rc == L.EINVAL and b"qnew must equal qold" in L.lib().pcl_last_error()
    L.lib().pcl_layer1_release()
