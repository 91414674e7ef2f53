"""
GPU (1 device): the device halo path -- pack kernel, RCCL group of Send/Recv, unpack kernel,
RCCL max all-reduce -- exercised on a single rank whose 8 neighbours are all itself (a fully
periodic 1 x 1 "decomposition").  The expected ghost frame is then numpy's periodic wrap.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_solver(L, mx, my, meqn_rp=(5, 5, 11), maux=0):
    cfg = L.Config()
    cfg.ndim = 2
    cfg.n[0], cfg.n[1] = mx, my
    cfg.mbc = 2
    cfg.meqn, cfg.mwaves, cfg.rp = meqn_rp
    cfg.maux = maux
    cfg.method[1] = 2
    cfg.method[2] = -1
    cfg.method[6] = maux
    for k in range(cfg.mwaves):
        cfg.mthlim[k] = 4
    cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.d[0], cfg.d[1] = 1.0 / mx, 1.0 / my
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    return h


@pytest.mark.parametrize("mx,my", [(8, 5), (70, 33)])
def test_self_periodic_halo(mx, my):
    from pyclaw_amd import _lib as L
    lib = L.lib()
    h = make_solver(L, mx, my, maux=2)
    try:
        uid = C.create_string_buffer(128)
        L.check(lib.pcl_comm_unique_id(uid))
        nbr = np.zeros(8, dtype=np.int32)          # every neighbour is rank 0
        L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(nbr)))
        rng = np.random.default_rng(1)
        g = 2
        qbc = np.asfortranarray(rng.standard_normal((5, mx + 2 * g, my + 2 * g)))
        L.check(lib.pcl_put_q(h, L.d(qbc), 1))
        L.check(lib.pcl_halo_exchange(h))
        out = np.zeros_like(qbc)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
        inner = qbc[:, g:-g, g:-g]
        expect = np.asfortranarray(np.pad(inner, ((0, 0), (g, g), (g, g)), mode="wrap"))
        assert np.array_equal(out, expect)
        # aux path uses the same machinery with maux components
        auxbc = np.asfortranarray(rng.standard_normal((2, mx + 2 * g, my + 2 * g)))
        L.check(lib.pcl_put_aux(h, L.d(auxbc)))
        L.check(lib.pcl_halo_exchange_aux(h))
        # all-reduce of one double
        v = C.c_double(0.375)
        L.check(lib.pcl_allreduce_max(h, C.cast(C.byref(v), L.dp)))
        assert v.value == 0.375
    finally:
        lib.pcl_destroy(h)


def test_partial_neighbours():
    """only E/W neighbours (x-periodic strip): y ghosts must stay untouched"""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    mx, my, g = 20, 9, 2
    h = make_solver(L, mx, my)
    try:
        uid = C.create_string_buffer(128)
        L.check(lib.pcl_comm_unique_id(uid))
        nbr = np.array([0, 0, -1, -1, -1, -1, -1, -1], dtype=np.int32)
        L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(nbr)))
        rng = np.random.default_rng(2)
        qbc = np.asfortranarray(rng.standard_normal((5, mx + 2 * g, my + 2 * g)))
        L.check(lib.pcl_put_q(h, L.d(qbc), 1))
        L.check(lib.pcl_halo_exchange(h))
        out = np.zeros_like(qbc)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
        expect = qbc.copy()
        expect[:, :g, g:-g] = qbc[:, -2 * g:-g, g:-g]
        expect[:, -g:, g:-g] = qbc[:, g:2 * g, g:-g]
        assert np.array_equal(out, expect)
    finally:
        lib.pcl_destroy(h)


def test_step_with_comm_single_rank():
    """pcl_step_hyperbolic on a solver that joined a (1-rank) communicator: the device-side CFL
    all-reduce must leave the step and its Courant number unchanged."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    mx, my, g = 64, 40, 2
    rng = np.random.default_rng(3)
    qbc = np.empty((5, mx + 2 * g, my + 2 * g), order="F")
    qbc[0] = 1 + 0.1 * rng.random(qbc.shape[1:])
    qbc[1] = 0.1 * rng.random(qbc.shape[1:])
    qbc[2] = 0.05 * rng.random(qbc.shape[1:])
    qbc[3] = 2.5 + 0.1 * rng.random(qbc.shape[1:])
    qbc[4] = rng.random(qbc.shape[1:])
    res = []
    for with_comm in (False, True):
        h = make_solver(L, mx, my)
        try:
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                nbr = np.full(8, -1, dtype=np.int32)
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(nbr)))
            L.check(lib.pcl_put_q(h, L.d(qbc), 1))
            cfl = C.c_double()
            L.check(lib.pcl_step_hyperbolic(h, 1e-3, C.cast(C.byref(cfl), L.dp)))
            out = np.zeros_like(qbc)
            L.check(lib.pcl_get_q(h, L.d(out), 1))
            res.append((out, cfl.value))
        finally:
            lib.pcl_destroy(h)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1] == res[1][1] and res[0][1] > 0


@pytest.mark.parametrize("mx,my,overlap", [(1000, 37, 1), (482, 64, 1), (723, 9, 1), (100, 30, 1), (1000, 37, 0), (1000, 37, 2),
                                               (481, 10, 2), (962, 11, 2),
                                               # the block of BASELINE configs[3]'s 2 x 4 layout (8192^2 / (2 x 4))
                                               (4096, 2048, 1), (4096, 2048, 0), (4096, 2048, 2)])
def test_overlapped_step_equals_periodic(mx, my, overlap, monkeypatch):
    """pcl_bc_step on a block whose 8 neighbours are itself (halo exchange on its own stream, overlapped
    with the interior tiles of the x pass; rim tiles afterwards) == the same steps with local periodic
    BCs, bit for bit, Courant number included.  (100, 30) has no interior box -> sequential fallback;
    overlap=0 forces the sequential path; overlap=2 is the deterministic race check: interior tiles are
    launched strictly BEFORE the exchange, so any ghost cell they read would be stale (NaN-poisoned here)."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", str(overlap))
    g = 2
    rng = np.random.default_rng(11)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    res = []
    for with_comm in (False, True):
        h = make_solver(L, mx, my)
        try:
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                bc = np.full(4, -1, dtype=np.int32)
            else:
                bc = np.full(4, 2, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")   # ghost frame starts as NaN
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            consts = np.zeros(4 * 8)
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), 2e-4 * 100 / mx, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and 0 < res[0][1][0] < 1
    assert np.array_equal(res[0][0], res[1][0])
    assert not np.array_equal(res[0][0], q0)


def test_sequential_decomposed_step_runs_both_forms(monkeypatch):
    """The default of a decomposed run whose blocks can all run the one-kernel step: exchange in front of the step, then
    the step in its FASTER form -- one kernel or x pass + y pass, re-measured by every rank for itself (trial steps 64
    steps into a window of 256; pcl_bc_step -> pcl_step_hyperbolic).  80 steps on a block whose 8 neighbours are itself,
    a rejected step and its retake among the trial steps: equal to the same sequence with local periodic fills, bit for
    bit, Courant numbers included, and both forms did run (pcl_step_form_stats)."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", "1")
    mx, my, g = 300, 200, 2
    rng = np.random.default_rng(31)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    q0[:, 40:170, 30:150] = q0[:, 40:41, 30:31]           # an undisturbed patch: tiles whose wavefronts take the shortcut
    dt = 2e-4 * 100 / mx
    res = []
    for with_comm in (False, True):
        h = make_solver(L, mx, my)
        try:
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                yes = C.c_int(0)
                L.check(lib.pcl_halo_can_overlap(h, C.byref(yes)))
                assert yes.value == 2                  # interior box of one-kernel tiles; no exchange-ahead is set
                bc = np.full(4, -1, dtype=np.int32)
            else:
                bc = np.full(4, 2, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            consts = np.zeros(4 * 8)
            cfls = []

            def step(d):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), d, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            for k in range(80):
                if k in (66, 70):                     # "rejected" among the trial steps of either form
                    step(3 * dt)
                    L.check(lib.pcl_undo_step(h))
                    step(0.7 * dt)
                else:
                    step(dt)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            ms = np.zeros(3)
            nl = np.zeros(3, dtype=np.int64)
            s1, s0 = C.c_long(0), C.c_long(0)
            L.check(lib.pcl_step_form_stats(h, L.d(ms), nl.ctypes.data_as(C.POINTER(C.c_long)), C.byref(s1), C.byref(s0)))
            res.append((out, cfls, (s1.value, s0.value)))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and 0 < res[0][1][0] < 1
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()
    for (_, _, (one, two)) in res:                     # 82 steps incl. the two rejected ones, both forms among them
        assert one + two == 82 and one >= 4 and two >= 4, res[0][2:] + res[1][2:]


@pytest.mark.parametrize("mx,my,order", [(1000, 37, 0), (723, 9, 0), (4096, 2048, 0), (4096, 2048, 1), (1024, 300, 1)])
def test_exchange_ahead_equals_periodic_with_retake(mx, my, order, monkeypatch):
    """pcl_halo_exchange_ahead: the halo of the new state is exchanged behind the y pass that produced it and the next
    step skips its own exchange.  A sequence with everything that can come between two steps -- accepted steps, a
    rejected one (pcl_undo_step: back to the pre-step buffer and its still-filled ghost frame), a retake with a smaller
    dt, an upload in between (the library must exchange again) -- equals the same sequence with local periodic fills,
    bit for bit, Courant numbers included.  The ghost frame starts NaN-poisoned."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", "1")
    g = 2
    rng = np.random.default_rng(23)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    dt = 2e-4 * 100 / mx
    res = []
    for with_comm in (False, True):
        h = make_solver(L, mx, my)
        try:
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                yes = C.c_int(0)
                L.check(lib.pcl_halo_can_overlap(h, C.byref(yes)))
                assert yes.value in (1, 2)       # 2: the one-kernel step with tile subsets (its own exchange-ahead order)
                # order 0: what the block offers; 1: the two-pass order although the block could run the one-kernel step
                # (what a run does when another rank's block is too thin for one-kernel tiles)
                L.check(lib.pcl_halo_exchange_ahead(h, 1 if order else yes.value))
                bc = np.full(4, -1, dtype=np.int32)
            else:
                bc = np.full(4, 2, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            consts = np.zeros(4 * 8)
            cfls = []

            def step(d):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), d, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            step(dt)
            step(dt)
            step(3 * dt)                       # "rejected": undo, retake with a smaller step
            L.check(lib.pcl_undo_step(h))
            step(0.7 * dt)
            step(dt)
            mid = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(mid), 0))
            L.check(lib.pcl_put_q(h, L.d(np.asfortranarray(mid[:, ::-1, :])), 0))    # new data: must be exchanged again
            step(dt)
            step(dt)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and 0 < res[0][1][0] < 1
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()


def test_exchange_ahead_refused_without_overlap():
    """no communicator / a block without interior x-pass tiles: the call says so instead of silently doing nothing"""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    h = make_solver(L, 100, 30)
    try:
        assert lib.pcl_halo_exchange_ahead(h, 1) != 0 and b"pcl_halo_exchange_ahead" in lib.pcl_last_error()
        uid = C.create_string_buffer(128)
        L.check(lib.pcl_comm_unique_id(uid))
        L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
        assert lib.pcl_halo_exchange_ahead(h, 1) != 0          # 100 x 30: no interior tile
        L.check(lib.pcl_halo_exchange_ahead(h, 0))
    finally:
        lib.pcl_destroy(h)


@pytest.mark.parametrize("shape", [(20, 9, 7), (70, 33, 18)])
def test_3d_self_halo_equals_periodic(shape):
    """3-D block cut in (y, z): the exchanged plane is (j, k) with whole x-rows as elements.  One rank whose 8
    neighbours are itself == local periodic ghost fills in y and z, bit for bit, over 3 dim-split steps."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    rng = np.random.default_rng(21)
    q0 = np.asfortranarray(rng.standard_normal((4,) + shape))
    aux = np.empty((2,) + tuple(n + 4 for n in shape), order="F")
    aux[0] = 1.0 + rng.random(aux.shape[1:])
    aux[1] = 0.5 + rng.random(aux.shape[1:])
    res = []
    for with_comm in (False, True):
        cfg = L.Config()
        cfg.ndim = 3
        for k in range(3):
            cfg.n[k] = shape[k]
            cfg.d[k] = 0.1
        cfg.mbc, cfg.meqn, cfg.mwaves, cfg.rp, cfg.maux = 2, 4, 2, 20, 2
        cfg.method[1], cfg.method[2], cfg.method[6] = 2, -1, 2
        cfg.mthlim[0] = cfg.mthlim[1] = 4
        h = C.c_void_p()
        L.check(lib.pcl_create(C.byref(cfg), C.byref(h)))
        try:
            L.check(lib.pcl_put_aux(h, L.d(aux)))
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                L.check(lib.pcl_halo_exchange_aux(h))
                bc = np.array([2, 2, -1, -1, -1, -1], dtype=np.int32)
            else:
                for idim in (1, 2):
                    for side in (0, 1):
                        L.check(lib.pcl_bc_aux(h, idim, side, 2))
                bc = np.full(6, 2, dtype=np.int32)
            for side in (0, 1):
                L.check(lib.pcl_bc_aux(h, 0, side, 2))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(48)), 0.01, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and res[0][1][0] > 0
    assert np.array_equal(res[0][0], res[1][0])
    assert not np.array_equal(res[0][0], q0)


@pytest.mark.parametrize("overlap", [1, 2])
@pytest.mark.parametrize("axis", ["x", "y"])
def test_overlapped_step_partial_neighbours(axis, overlap, monkeypatch):
    """Neighbours on one axis only (self, periodic strip); the other axis has physical BCs (reflecting lower,
    outflow upper) evaluated inside the x pass.  The interior box then reaches the sides without a neighbour.
    == the same steps without a communicator and local periodic fills on that axis, bit for bit; overlap=2 is
    the deterministic race check over a NaN-poisoned ghost frame."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", str(overlap))
    mx, my, g = 1210, 41, 2
    rng = np.random.default_rng(5)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    res = []
    for with_comm in (False, True):
        h = make_solver(L, mx, my)
        try:
            if axis == "x":
                nbr = np.array([0, 0, -1, -1, -1, -1, -1, -1], dtype=np.int32)
                bc_comm, bc_loc = [-1, -1, 3, 1], [2, 2, 3, 1]
            else:
                nbr = np.array([-1, -1, 0, 0, -1, -1, -1, -1], dtype=np.int32)
                bc_comm, bc_loc = [3, 1, -1, -1], [3, 1, 2, 2]
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(nbr)))
            bc = np.array(bc_comm if with_comm else bc_loc, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(32)), 2e-5, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and res[0][1][0] > 0
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()


def make_unsplit_solver(L, mx, my, capa):
    cfg = L.Config()
    cfg.ndim = 2
    cfg.n[0], cfg.n[1] = mx, my
    cfg.mbc = 2
    cfg.meqn, cfg.mwaves, cfg.rp = 5, 5, 11
    cfg.maux = 1 if capa else 0
    cfg.method[1] = 2
    cfg.method[2] = 2            # unsplit, transverse increment + correction waves
    cfg.method[5] = 1 if capa else 0
    cfg.method[6] = cfg.maux
    for k in range(cfg.mwaves):
        cfg.mthlim[k] = 4
    cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.d[0], cfg.d[1] = 1.0 / mx, 1.0 / my
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    return h


@pytest.mark.parametrize("mx,my,overlap,capa", [(300, 47, 1, False), (300, 47, 2, False), (300, 47, 0, False),
                                                    (190, 31, 2, True), (190, 31, 1, True), (61, 13, 2, False),
                                                    (125, 16, 2, False), (4096, 2048, 1, False), (4096, 2048, 2, False)])
def test_overlapped_unsplit_step_equals_periodic(mx, my, overlap, capa, monkeypatch):
    """The unsplit step (step2.f) of a block whose 8 neighbours are itself: ghost frame built on the halo stream while
    the x phase runs the tiles that read no ghost cell, rim tiles behind it, y phase after the join == the same steps
    with local periodic fills, bit for bit.  overlap=2: interior tiles strictly BEFORE the exchange over a NaN-poisoned
    ghost frame (race check); overlap=0: sequential exchange; (61, 13) has no interior tile at all."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", str(overlap))
    g = 2
    rng = np.random.default_rng(23)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    aux = np.asfortranarray(np.pad(1.0 + 0.3 * rng.random((1, mx, my)), ((0, 0), (g, g), (g, g)), mode="wrap"))
    res = []
    for with_comm in (False, True):
        h = make_unsplit_solver(L, mx, my, capa)
        try:
            if capa:
                L.check(lib.pcl_put_aux(h, L.d(aux)))
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                bc = np.full(4, -1, dtype=np.int32)
            else:
                bc = np.full(4, 2, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(32)), 2e-4 * 100 / mx, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and 0 < res[0][1][0] < 1
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()
    assert not np.array_equal(res[0][0], q0)


@pytest.mark.parametrize("overlap", [1, 2])
@pytest.mark.parametrize("axis", ["x", "y"])
def test_overlapped_unsplit_partial_neighbours(axis, overlap, monkeypatch):
    """Unsplit step with neighbours on one axis only (self, periodic strip) and physical BCs (reflecting lower,
    outflow upper) on the other: the BC fills run on the halo stream behind the exchange (corner ghosts read
    exchanged cells).  == local periodic fills on that axis, bit for bit."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", str(overlap))
    mx, my, g = 250, 41, 2
    rng = np.random.default_rng(6)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    res = []
    for with_comm in (False, True):
        h = make_unsplit_solver(L, mx, my, False)
        try:
            if axis == "x":
                nbr = np.array([0, 0, -1, -1, -1, -1, -1, -1], dtype=np.int32)
                bc_comm, bc_loc = [-1, -1, 3, 1], [2, 2, 3, 1]
            else:
                nbr = np.array([-1, -1, 0, 0, -1, -1, -1, -1], dtype=np.int32)
                bc_comm, bc_loc = [3, 1, -1, -1], [3, 1, 2, 2]
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(nbr)))
            bc = np.array(bc_comm if with_comm else bc_loc, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                L.check(lib.pcl_bc_step(h, L.i(bc), L.d(np.zeros(32)), 1e-4, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and res[0][1][0] > 0
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()


def make_sharp_solver(L, mx, my, lim=2):
    cfg = L.Config()
    cfg.ndim = 2
    cfg.n[0], cfg.n[1] = mx, my
    cfg.d[0], cfg.d[1] = 1.0 / mx, 1.0 / my
    cfg.mbc = 3
    cfg.meqn, cfg.mwaves, cfg.rp = 5, 5, 11
    cfg.method[1] = 2
    cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.kind = 1
    cfg.lim_type = lim
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    return h


@pytest.mark.parametrize("mx,my,overlap", [(300, 70, 1), (300, 70, 2), (300, 70, 0), (140, 33, 2), (60, 20, 2)])
@pytest.mark.parametrize("fused", [True, False])
def test_overlapped_sharpclaw_stage_equals_periodic(mx, my, overlap, fused, monkeypatch):
    """SharpClaw stages (frame + flux2 + forward-Euler combination q += dq, three in a row) on a block whose 8
    neighbours are itself: the stage's ghost frame is built on the halo stream while the x pass runs the tiles that
    read no ghost cell == the same stages with local periodic fills, bit for bit.  overlap=2: interior tiles strictly
    BEFORE the exchange over a NaN-poisoned ghost frame; (60, 20) has no interior tile.  fused=False: pcl_sharp_bc_dq
    + a separate register operation."""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    monkeypatch.setenv("PCL_HALO_OVERLAP", str(overlap))
    g = 3
    rng = np.random.default_rng(31)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    res = []
    for with_comm in (False, True):
        h = make_sharp_solver(L, mx, my)
        try:
            if with_comm:
                uid = C.create_string_buffer(128)
                L.check(lib.pcl_comm_unique_id(uid))
                L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
                bc = np.full(4, -1, dtype=np.int32)
            else:
                bc = np.full(4, 2, dtype=np.int32)
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            dt = 2e-4 * 100 / mx
            for _ in range(3):
                cfl = C.c_double()
                if fused:
                    L.check(lib.pcl_sharp_bc_stage(h, L.i(bc), L.d(np.zeros(32)), dt, 1, 0, 0, 0, 1.0, 0.0, 0.0, 1e9,
                                                   C.cast(C.byref(cfl), L.dp)))
                else:
                    L.check(lib.pcl_sharp_bc_dq(h, L.i(bc), L.d(np.zeros(32)), dt, C.cast(C.byref(cfl), L.dp)))
                    L.check(lib.pcl_rk_op(h, 1, 0, 0, 3, 0, 1.0, 0.0, 0.0))       # q = q + dq/1
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and 0 < res[0][1][0] < 1
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()
    assert not np.array_equal(res[0][0], q0)


def test_unsplit_frame_constant_state_bc():
    """pcl_bc_step on an unsplit solver: the one-launch ghost frame with a CONSTANT-STATE side (the shock-bubble inflow)
    over a NaN-poisoned frame == ghost fills one by one (pcl_bc_const / pcl_bc) + pcl_step_hyperbolic.  (A constant
    cell maps to itself in the frame's index remap: an early version skipped writing it to q.  Found by the
    multi-process run of tests/test_gpu_multiproc.py.)"""
    from pyclaw_amd import _lib as L
    lib = L.lib()
    mx, my, g = 150, 37, 2
    rng = np.random.default_rng(77)
    q0 = np.empty((5, mx, my), order="F")
    q0[0] = 1 + 0.3 * rng.random((mx, my))
    q0[1] = 0.3 * rng.standard_normal((mx, my))
    q0[2] = 0.2 * rng.standard_normal((mx, my))
    q0[3] = 2.5 + 0.5 * rng.random((mx, my))
    q0[4] = rng.random((mx, my))
    inflow = np.array([1.3, 0.4, 0.0, 3.1, 0.0])
    bc = np.array([0, 1, 3, 1], dtype=np.int32)             # custom (constant), outflow, reflecting, outflow
    consts = np.zeros(32)
    consts[0:5] = inflow
    res = []
    for fused in (True, False):
        h = make_unsplit_solver(L, mx, my, False)
        try:
            poison = np.full((5, mx + 2 * g, my + 2 * g), np.nan, order="F")
            L.check(lib.pcl_put_q(h, L.d(poison), 1))
            L.check(lib.pcl_put_q(h, L.d(q0), 0))
            cfls = []
            for _ in range(3):
                cfl = C.c_double()
                if fused:
                    L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), 1e-4, C.cast(C.byref(cfl), L.dp)))
                else:
                    L.check(lib.pcl_bc_const(h, 0, 0, L.d(inflow)))
                    L.check(lib.pcl_bc(h, 0, 1, 1))
                    L.check(lib.pcl_bc(h, 1, 0, 3))
                    L.check(lib.pcl_bc(h, 1, 1, 1))
                    L.check(lib.pcl_step_hyperbolic(h, 1e-4, C.cast(C.byref(cfl), L.dp)))
                cfls.append(cfl.value)
            out = np.zeros_like(q0)
            L.check(lib.pcl_get_q(h, L.d(out), 0))
            res.append((out, cfls))
        finally:
            lib.pcl_destroy(h)
    assert res[0][1] == res[1][1] and res[0][1][0] > 0
    assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all()
