"""
GPU, BASELINE.json's full sizes (-m gpu): where the oracle cannot replay a whole grid in seconds, parity is
checked through (a) SPOT PARITY -- windows cut out of the full-size input are stepped by the CPU oracle and
compared bit for bit with the same cells of the full-size GPU result (the stencil reach tells how much
context a window needs), and (b) TRANSLATION INVARIANCE -- on a periodic grid, stepping a cyclically shifted
state equals shifting the stepped state, bit for bit and with the same Courant number (every cell meets
different tile / strip / wavefront boundaries in the two runs).

  C3  apps/euler 2D shock-bubble physics, 4096 x 4096, classic dim-split          (headline configuration)
  C2  apps/acoustics 2D, 1024 x 1024, classic unsplit (rpn2/rpt2) + MC limiter
  C5* SharpClaw WENO5 right-hand side, Euler, 2048 x 1024                          (the C5 grid; RS = Euler)
  3-D acoustics 256^3, classic dim-split
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import oracle as O


def make(L, ndim, n, rp, meqn, mwaves, mthlim, method2, maux=0, kind=0, mbc=2, params=(1.4, 0.4), d=None):
    cfg = L.Config()
    cfg.ndim = ndim
    for k in range(ndim):
        cfg.n[k] = n[k]
        cfg.d[k] = (d or [1.0 / n[k]] * ndim)[k] if d else 1.0 / n[k]
    cfg.mbc = mbc
    cfg.meqn, cfg.mwaves, cfg.rp, cfg.maux = meqn, mwaves, rp, maux
    cfg.method[1] = 2
    cfg.method[2] = method2
    cfg.method[6] = maux
    for k, v in enumerate(mthlim):
        cfg.mthlim[k] = v
    for k, v in enumerate(params):
        cfg.rp_params[k] = v
    cfg.kind = kind
    cfg.lim_type = 2
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    return h


def euler_field(rng, shape):
    """smooth random Euler state with a few strong jumps and constant patches"""
    q = np.empty((5,) + shape, order="F")
    q[0] = 1.0 + 0.1 * rng.random(shape)
    q[1] = 0.1 * rng.random(shape) - 0.03
    q[2] = 0.05 * rng.random(shape) - 0.02
    q[3] = 2.5 + 0.1 * rng.random(shape)
    q[4] = rng.random(shape)
    nx, ny = shape
    q[:, nx // 3:nx // 3 + nx // 8, ny // 5:ny // 5 + ny // 4] = np.array([0.3, 0.2, -0.1, 1.1, 1.0])[:, None, None]
    q[:, nx // 2:, :ny // 7] *= np.array([2.0, 2.0, 2.0, 3.0, 1.0])[:, None, None]
    return q


def windows(rng, shape, w, pad, count):
    for _ in range(count):
        yield tuple(int(rng.integers(pad, n - w - pad)) for n in shape)


@pytest.mark.parametrize("n", [4096, 8192])      # C3's grid and C4's global grid (8192^2: 2.7 GB per buffer)
def test_c3_euler_4096_spot_parity_and_shift(n, coracle):
    from pyclaw_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(42)
    mth = [4, 4, 4, 4, 2]
    dt = 0.1 / n
    cflp = C.c_double()
    # ---- spot parity: ghost cells supplied, one dim-split step (x sweep then y sweep, clawpack.py:538-546)
    qbc = euler_field(rng, (n + 4, n + 4))
    h = make(L, 2, (n, n), O.RP_EULER5_2D, 5, 5, mth, -1)
    try:
        L.check(lib.pcl_put_q(h, L.d(qbc), 1))
        L.check(lib.pcl_step_hyperbolic(h, dt, C.cast(C.byref(cflp), L.dp)))
        out = np.empty_like(qbc)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
    finally:
        lib.pcl_destroy(h)
    assert 0 < cflp.value < 1 and np.isfinite(out).all()
    method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
    w = 48
    corners = list(windows(rng, (n + 4, n + 4), w, 8, 6)) + [(4, 4), (n + 4 - w - 6, 4), (4, n + 4 - w - 6),
                                                             (n // 3 - 10, n // 5 - 10)]
    for (i0, j0) in corners:
        # block: w x (w+4) interior cells + 2 ghost layers; after x then y sweep its rows 2.. are exact (stencil 2)
        blk = np.array(qbc[:, i0 - 2:i0 + w + 2, j0 - 4:j0 + w + 4], order="F")
        ref = blk.copy("F")
        coracle.step2ds(O.RP_EULER5_2D, [1.4, 0.4], w + 4, 2, w, w + 4, blk, ref, None, 1.0 / n, 1.0 / n, dt,
                        method, mth, 1)
        coracle.step2ds(O.RP_EULER5_2D, [1.4, 0.4], w + 4, 2, w, w + 4, ref, ref, None, 1.0 / n, 1.0 / n, dt,
                        method, mth, 2)
        assert np.array_equal(out[:, i0:i0 + w, j0:j0 + w], ref[:, 2:-2, 4:-4]), (i0, j0)
        assert ref[:, 2:-2, 4:-4].shape == (5, w, w) and not np.array_equal(ref[:, 2:-2, 4:-4], blk[:, 2:-2, 4:-4])
    # ---- translation invariance on the periodic grid
    q0 = np.array(qbc[:, 2:-2, 2:-2], order="F")
    del qbc, out
    bc = np.full(4, 2, dtype=np.int32)
    res = []
    for shift in ((0, 0), (37, 1001)):
        qs = np.asfortranarray(np.roll(q0, shift, axis=(1, 2)))
        h = make(L, 2, (n, n), O.RP_EULER5_2D, 5, 5, mth, -1)
        try:
            L.check(lib.pcl_put_q(h, L.d(qs), 0))
            cfls = []
            for _ in range(2):
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(32)), dt, C.cast(C.byref(cflp), L.dp)))
                cfls.append(cflp.value)
            L.check(lib.pcl_get_q(h, L.d(qs), 0))
        finally:
            lib.pcl_destroy(h)
        res.append((np.roll(qs, (-shift[0], -shift[1]), axis=(1, 2)), cfls))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])
    # conservation on the periodic grid: mass, momenta and energy move only by rounding (the tracer of the
    # 5-wave solver is advected in non-conservative form, rpn2_euler_5wave.f:160-163)
    for m in range(4):
        s0, s1 = q0[m].sum(dtype=np.longdouble), res[0][0][m].sum(dtype=np.longdouble)
        assert abs(s1 - s0) <= 1e-12 * np.abs(q0[m]).sum(dtype=np.longdouble), m


def test_c2_acoustics_1024_unsplit_spot_parity_and_shift(coracle):
    from pyclaw_amd import _lib as L
    lib = L.lib()
    n = 1024
    rng = np.random.default_rng(7)
    par = [1.0, 4.0, 2.0, 2.0]
    mth = [4, 4]
    dt = 0.2 / n
    d = 2.0 / n
    cflp = C.c_double()
    qbc = np.asfortranarray(rng.standard_normal((3, n + 4, n + 4)))
    qbc[:, 300:500, 100:400] = 0.25
    method = np.array([1, 2, 2, 0, 0, 0, 0], dtype=np.int32)
    h = make(L, 2, (n, n), O.RP_ACOUSTICS_2D, 3, 2, mth, 2, params=par, d=[d, d])
    try:
        L.check(lib.pcl_put_q(h, L.d(qbc), 1))
        L.check(lib.pcl_step_hyperbolic(h, dt, C.cast(C.byref(cflp), L.dp)))
        out = np.empty_like(qbc)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
    finally:
        lib.pcl_destroy(h)
    w = 40
    for (i0, j0) in list(windows(rng, (n + 4, n + 4), w + 8, 8, 6)) + [(290, 90)]:
        # unsplit step: cell (i,j) sees slices j-1..j+1 / i-1..i+1 each with reach 2 => context 3; block = w+8 cells + ghosts
        blk = np.array(qbc[:, i0 - 2:i0 + w + 10, j0 - 2:j0 + w + 10], order="F")
        ref = blk.copy("F")
        coracle.step2(O.RP_ACOUSTICS_2D, par, w + 8, 2, w + 8, w + 8, blk, ref, None, d, d, dt, method, mth)
        assert np.array_equal(out[:, i0 + 4:i0 + 4 + w, j0 + 4:j0 + 4 + w], ref[:, 6:-6, 6:-6]), (i0, j0)
        assert ref[:, 6:-6, 6:-6].shape == (3, w, w)
    q0 = np.array(qbc[:, 2:-2, 2:-2], order="F")
    bc = np.full(4, 2, dtype=np.int32)
    res = []
    for shift in ((0, 0), (513, 77)):
        qs = np.asfortranarray(np.roll(q0, shift, axis=(1, 2)))
        h = make(L, 2, (n, n), O.RP_ACOUSTICS_2D, 3, 2, mth, 2, params=par, d=[d, d])
        try:
            L.check(lib.pcl_put_q(h, L.d(qs), 0))
            for _ in range(3):
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(32)), dt, C.cast(C.byref(cflp), L.dp)))
            L.check(lib.pcl_get_q(h, L.d(qs), 0))
        finally:
            lib.pcl_destroy(h)
        res.append((np.roll(qs, (-shift[0], -shift[1]), axis=(1, 2)), cflp.value))
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])


def test_sharpclaw_capa_euler_2048x1024_spot_parity(coracle):
    """SharpClaw WENO5 right-hand side (flux2.f90) with a capacity function at the C5 grid size, Euler solver.  (The C5
    configuration itself -- the sphere solver with its 16 aux components -- is tests/test_gpu_sphere.py.)"""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    nx, ny = 2048, 1024
    rng = np.random.default_rng(3)
    q = euler_field(rng, (nx + 6, ny + 6))
    aux = np.asfortranarray(0.5 + rng.random((1, nx + 6, ny + 6)))
    dx, dy, dt = 1.0 / nx, 1.0 / ny, 0.05 / nx
    cflp = C.c_double()
    # capacity function: mcapa = 1 (first aux component)
    dq = np.zeros_like(q)
    L.check(lib.pcl_sharp_flux2(O.RP_EULER5_2D, L.d(np.array([1.4, 0.4] + [0.0] * 6)), 2, 5, 5, 1, 1, 3, nx, ny,
                                L.d(q), L.d(dq), L.d(aux), dx, dy, dt, C.cast(C.byref(cflp), L.dp)))
    assert np.isfinite(dq[:, 3:-3, 3:-3]).all() and cflp.value > 0 and np.abs(dq[:, 3:-3, 3:-3]).max() > 0
    w = 32
    for (i0, j0) in list(windows(rng, (nx + 6, ny + 6), w, 8, 5)) + [(3, 3), (nx + 3 - w, ny + 3 - w)]:
        qb = np.array(q[:, i0 - 3:i0 + w + 3, j0 - 3:j0 + w + 3], order="F")
        ab = np.array(aux[:, i0 - 3:i0 + w + 3, j0 - 3:j0 + w + 3], order="F")
        ref, _ = coracle.sharp_flux2(O.RP_EULER5_2D, [1.4, 0.4], 2, 5, 1, 3, w, w, qb, ab, dx, dy, dt)
        assert np.array_equal(dq[:, i0:i0 + w, j0:j0 + w], ref[:, 3:-3, 3:-3]), (i0, j0)
        assert ref[:, 3:-3, 3:-3].shape == (5, w, w)


def test_3d_acoustics_256_spot_parity_and_shift(coracle):
    from pyclaw_amd import _lib as L
    lib = L.lib()
    n = (256, 192, 160)
    rng = np.random.default_rng(11)
    full = tuple(k + 4 for k in n)
    q = np.asfortranarray(rng.standard_normal((4,) + full))
    q[:, 50:120, 30:90, 20:100] = 0.5
    aux = np.empty((2,) + full, order="F")
    aux[0] = 1.0 + (np.arange(full[0])[:, None, None] > full[0] // 2)
    aux[1] = 1.0 + 0.5 * (np.arange(full[1])[None, :, None] > full[1] // 3)
    d = (2.0 / n[0], 2.0 / n[1], 2.0 / n[2])
    dt = 0.3 * min(d) / 2.0
    method = np.array([1, 2, -1, 0, 0, 0, 2], dtype=np.int32)
    mth = np.array([4, 4], dtype=np.int32)
    cflp = C.c_double()
    h = make(L, 3, n, O.RP_VC_ACOUSTICS_3D, 4, 2, [4, 4], -1, maux=2, params=(), d=list(d))
    try:
        L.check(lib.pcl_put_q(h, L.d(q), 1))
        L.check(lib.pcl_put_aux(h, L.d(aux)))
        L.check(lib.pcl_step_hyperbolic(h, dt, C.cast(C.byref(cflp), L.dp)))
        out = np.empty_like(q)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
    finally:
        lib.pcl_destroy(h)
    assert 0 < cflp.value < 1
    w = 12
    for c in list(windows(rng, full, w + 8, 6, 4)) + [(44, 24, 14)]:
        # x, y, z sweeps in turn, reach 2 each: context 4 in y for x-sweep results used by y, etc. -> pad 4 + ghosts
        sl = tuple(slice(c[k] - 2, c[k] + w + 10) for k in range(3))
        qb = np.array(q[(slice(None),) + sl], order="F")
        ab = np.array(aux[(slice(None),) + sl], order="F")
        ref = qb.copy("F")
        m = w + 8
        coracle.step3ds(O.RP_VC_ACOUSTICS_3D, m, 2, m, m, m, qb, ref, ab, d[0], d[1], d[2], dt, method, mth, 1)
        coracle.step3ds(O.RP_VC_ACOUSTICS_3D, m, 2, m, m, m, ref, ref, ab, d[0], d[1], d[2], dt, method, mth, 2)
        coracle.step3ds(O.RP_VC_ACOUSTICS_3D, m, 2, m, m, m, ref, ref, ab, d[0], d[1], d[2], dt, method, mth, 3)
        inner = tuple(slice(c[k] + 4, c[k] + 4 + w) for k in range(3))
        assert np.array_equal(out[(slice(None),) + inner], ref[:, 6:-6, 6:-6, 6:-6]), c
        assert ref[:, 6:-6, 6:-6, 6:-6].shape == (4, w, w, w)
