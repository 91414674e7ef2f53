"""
CPU: host-side logic of the pyclaw_amd surface that does not need the device.
"""
import numpy as np
import pytest

import pyclaw_amd as pyclaw
from pyclaw_amd import parallel, riemann


def test_dimension_matches_reference_formulas():
    """grid.py:54-87: d=(upper-lower)/float(n), center[i]=lower+(i+0.5)*d, edge[i]=lower+i*d."""
    x = pyclaw.Dimension('x', -1.0, 2.0, 7)
    d = 3.0 / 7.0
    assert x.d == d and x.ng == 7 and x.nstart == 0 and x.nend == 7
    assert np.array_equal(x.center, np.array([-1.0 + (i + 0.5) * d for i in range(7)]))
    assert np.array_equal(x.edge, np.array([-1.0 + i * d for i in range(8)]))
    y = pyclaw.Dimension(0.0, 1.0, 4)
    assert y.name == 'x' and y.n == 4


def test_state_layout():
    g = pyclaw.Grid([pyclaw.Dimension('x', 0., 1., 5), pyclaw.Dimension('y', 0., 1., 3)])
    s = pyclaw.State(g, 4, 2)
    assert s.q.shape == (4, 5, 3) and s.q.flags.f_contiguous
    assert s.aux.shape == (2, 5, 3) and s.meqn == 4 and s.maux == 2 and s.mcapa == -1
    assert pyclaw.State(g, 1).aux is None
    qbc = np.zeros((4, 9, 7), order='F')
    s.q[...] = 1.5
    s.get_qbc_from_q(2, 'q', qbc)
    assert qbc[:, 2:-2, 2:-2].min() == 1.5 and qbc[:, :2].max() == 0.0
    s.set_q_from_qbc(2, qbc)
    assert s.q.base is qbc or s.q.base is qbc.base   # a VIEW, state.py:182


def test_solver_defaults_match_reference():
    """clawpack.py:94-108, solver.py:126-190"""
    s = pyclaw.ClawSolver2D()
    assert (s.mbc, s.order, s.src_split, s.fwave, s.dim_split, s.order_trans) == (2, 2, 1, False, True, 1)
    assert (s.cfl_max, s.cfl_desired, s.dt_initial, s.dt_max, s.max_steps, s.dt_variable) == \
        (1.0, 0.9, 0.1, 1e99, 1000, True)
    assert s.limiters == 1 and s.step_src is None and s.start_step is None
    assert s.bc_lower == [None, None] and s.status['numsteps'] == 0 and s.status['cflmax'] == 0.9
    assert (pyclaw.BC.custom, pyclaw.BC.outflow, pyclaw.BC.periodic, pyclaw.BC.reflecting) == (0, 1, 2, 3)
    assert pyclaw.ClawSolver2D.trans_cor == 2 and pyclaw.ClawSolver1D().ndim == 1


def test_set_mthlim_and_method():
    s = pyclaw.ClawSolver2D()
    s.mwaves = 5
    s.limiters = 4
    s.set_mthlim()
    assert s.mthlim == [4] * 5
    s.limiters = [4, 4, 4, 4, 2]
    s.set_mthlim()
    assert s.mthlim == [4, 4, 4, 4, 2]
    s.limiters = [1, 2]
    with pytest.raises(Exception, match="Length of solver.limiters"):
        s.set_mthlim()
    g = pyclaw.Grid([pyclaw.Dimension('x', 0., 1., 5), pyclaw.Dimension('y', 0., 1., 3)])
    st = pyclaw.State(g, 5, 3)
    st.mcapa = 1
    s.set_method(st)
    assert list(s.method) == [1, 2, -1, 0, 0, 2, 3]
    s.dim_split = False
    s.order_trans = 2
    s.set_method(st)
    assert s.method[2] == 2


def test_cparam_check():
    """state.py:156-160: every cparam name must be in aux_global"""
    with pytest.raises(Exception, match="cparam"):
        riemann.rp_euler_5wave_2d.params({'gamma': 1.4})
    assert riemann.rp_euler_5wave_2d.params({'gamma': 1.4, 'gamma1': 0.4, 'extra': 1}) == [1.4, 0.4]
    assert riemann.get('euler_5wave_2d') is riemann.rp_euler_5wave_2d
    assert riemann.get('rp_acoustics_2d') is riemann.rp_acoustics_2d


def test_evolve_errors_without_setup():
    s = pyclaw.ClawSolver1D()
    g = pyclaw.Grid(pyclaw.Dimension('x', 0., 1., 10))
    sol = pyclaw.Solution(pyclaw.State(g, 1))
    with pytest.raises(Exception, match="setup"):
        s.evolve_to_time(sol, 1.0)


def test_proc_grid_rule():
    """PETSc DMDA default: 8192^2 on 8 ranks -> 2 x 4 (BASELINE config 4)."""
    assert parallel.proc_grid([8192, 8192], 8) == [2, 4]
    assert parallel.proc_grid([4096, 4096], 4) == [2, 2]
    assert parallel.proc_grid([4096, 4096], 2) == [1, 2]
    assert parallel.proc_grid([4096, 1024], 4) == [4, 1]
    assert parallel.proc_grid([100, 100], 6) == [2, 3]
    assert parallel.proc_grid([1000], 4) == [4]


def test_block_ranges_cover():
    for n, p in [(10, 3), (4096, 4), (7, 7), (100, 6)]:
        r = [parallel.block_range(n, p, c) for c in range(p)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(p - 1))
        sizes = [b - a for a, b in r]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_neighbors():
    W, E, S, N, SW, SE, NW, NE = range(8)
    d = parallel.Decomposition([64, 64], 8, 0)      # 2 x 4, rank 0 = (0,0)
    assert d.dims == [2, 4] and d.coords == [0, 0]
    n = d.neighbors([False, False])
    assert n[W] == -1 and n[S] == -1 and n[E] == 1 and n[N] == 2 and n[NE] == 3 and n[SW] == -1
    n = d.neighbors([True, True])
    assert n[W] == 1 and n[E] == 1 and n[S] == 6 and n[N] == 2 and n[SW] == 7 and n[NE] == 3
    d = parallel.Decomposition([64, 64], 8, 5)      # (1,2)
    assert d.coords == [1, 2]
    n = d.neighbors([False, False])
    assert n == [4, -1, 3, 7, 2, -1, 6, -1]
    # a dimension held by one block never exchanges, even when periodic (local BC copy instead)
    d = parallel.Decomposition([64, 16], 2, 1)
    assert d.dims == [2, 1]
    n = d.neighbors([True, True])
    assert n[S] == -1 and n[N] == -1 and n[SW] == -1 and n[W] == 0 and n[E] == 0
    # symmetry: if b is a's neighbour towards d, a is b's neighbour towards opposite(d)
    opp = [1, 0, 3, 2, 7, 6, 5, 4]
    for per in ([False, False], [True, False], [True, True]):
        decs = [parallel.Decomposition([60, 90], 6, r) for r in range(6)]
        for a in decs:
            na = a.neighbors(per)
            for dd in range(8):
                if na[dd] >= 0:
                    assert decs[na[dd]].neighbors(per)[opp[dd]] == a.rank


def test_riemann_registry_and_limiter_ids():
    import pyclaw_amd as pyclaw
    from pyclaw_amd import riemann
    ids = [r.id for r in riemann._ALL]
    assert len(ids) == len(set(ids)) == 16
    assert riemann.get('rp_euler_5wave_2d') is riemann.rp_euler_5wave_2d and riemann.get('burgers_1d').meqn == 1
    assert riemann.rp_vc_acoustics_2d.has_transverse and riemann.rp_shallow_2d.cparam == ('g',)
    with pytest.raises(Exception):
        riemann.get('no_such_solver')
    with pytest.raises(Exception):                       # cparam keys must be in aux_global (state.py:156-160)
        riemann.rp_euler_1d.params({'gamma': 1.4})
    t = pyclaw.limiters.tvd
    assert (t.minmod, t.superbee, t.vanleer, t.MC) == (1, 2, 3, 4)


def test_3d_decomposition_cuts_y_and_z_only():
    from pyclaw_amd import parallel
    for rank in range(8):
        d = parallel.Decomposition([64, 48, 40], 8, rank)
        assert d.axes == [1, 2] and d.ranges[0] == (0, 64)
        assert d.dims[0] * d.dims[1] == 8
    cover = np.zeros((48, 40), dtype=int)
    for rank in range(8):
        d = parallel.Decomposition([64, 48, 40], 8, rank)
        cover[d.ranges[1][0]:d.ranges[1][1], d.ranges[2][0]:d.ranges[2][1]] += 1
    assert (cover == 1).all()
    d = parallel.Decomposition([64, 48, 40], 8, 0)
    nb = d.neighbors([True, True, False])               # periodic in x (not cut) and y, not in z
    assert nb[parallel.W] >= 0 or d.dims[0] == 1
    assert nb[parallel.S] == -1                           # rank 0 sits at the lower z edge, z is not periodic


def test_clawsolver3d_surface():
    import pyclaw_amd as pyclaw
    s = pyclaw.ClawSolver3D()
    assert s.ndim == 3 and s.dim_split is True and s.order_trans == 22
    assert (s.no_trans, s.trans_inc, s.trans_cor) == (0, 11, 22)
    assert len(s.bc_lower) == 3


def test_comm_init_arguments_are_validated_before_rccl():
    """pcl_comm_check: what pcl_comm_init runs first (host code, no GPU): misuse gets a message, not RCCL's bare
    "invalid usage" (round-1 record gpurun_out/mgpu.log)."""
    import numpy as np
    from pyclaw_amd import _lib as L
    lib = L.lib()
    nb = lambda *v: L.i(np.array(v, dtype=np.int32))
    free = [-1] * 8
    assert lib.pcl_comm_check(2, 0, nb(*free)) == 0
    assert lib.pcl_comm_check(2, 2, nb(*free)) == L.EINVAL and b"rank 2 outside 0..1" in lib.pcl_last_error()
    assert lib.pcl_comm_check(0, 0, nb(*free)) == L.EINVAL
    assert lib.pcl_comm_check(4, 1, nb(0, 5, -1, -1, -1, -1, -1, -1)) == L.EINVAL
    assert b"neighbour E = 5" in lib.pcl_last_error()
    # a block may be its own neighbour only across a periodic dimension it spans alone: on BOTH faces
    assert lib.pcl_comm_check(2, 1, nb(0, 0, 1, 1, 0, 0, 0, 0)) == 0          # 2 x 1 blocks, periodic in x and y
    assert lib.pcl_comm_check(2, 1, nb(0, 0, 1, -1, -1, -1, -1, -1)) == L.EINVAL
    assert b"S/N" in lib.pcl_last_error()


def test_package_control_plane_is_torch_free():
    """north_star: no PyTorch.  The rendezvous / barrier / host reductions are pyclaw_amd.parallel's own TCP group."""
    import subprocess
    import sys
    code = ("import sys; import pyclaw_amd; from pyclaw_amd import parallel; parallel.init(); "
            "assert 'torch' not in sys.modules, 'torch imported'; print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT if 'ROOT' in globals() else None)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr


def test_1d_grids_are_decomposed_like_petclaw():
    """petclaw/state.py:199-234 cuts 1-D grids too: ranges (remainder first, PETSc's rule) and W / E neighbours with the
    periodic wrap; no S / N / corner neighbours"""
    from pyclaw_amd import parallel
    for size in (2, 3, 5):
        got = []
        for r in range(size):
            dec = parallel.Decomposition([1000], size, r)
            assert dec.dims == [size] and dec.coords == [r]
            got.append(dec.ranges[0])
            nb = dec.neighbors([True])
            assert nb[2:] == [-1] * 6
            assert nb[0] == (r - 1) % size and nb[1] == (r + 1) % size
            nb = dec.neighbors([False])
            assert nb[0] == (r - 1 if r > 0 else -1) and nb[1] == (r + 1 if r < size - 1 else -1)
        assert got[0][0] == 0 and got[-1][1] == 1000 and all(got[k][1] == got[k + 1][0] for k in range(size - 1))
        assert max(b - a for a, b in got) - min(b - a for a, b in got) <= 1
