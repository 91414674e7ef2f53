"""
GPU end-to-end replays (run with -m gpu) of the reference's regression scripts through the
pyclaw_amd solver surface, checked against the reference's own golden files and against the
oracle driver (same accept/reject sequence, same dt history, same final array).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import driver as D
from apps import problems


def test_shockbubble_golden(coracle, golden_dir):
    """test/test_examples.py:385-397: 160x40, t=0.2, density vs test/sb_density, gate 1e-12.
    We require bit equality (the oracle and the flang build both give 0.0)."""
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw)
    dens = claw.frames[claw.nout].state.q[0, :, :]
    gold = np.loadtxt(os.path.join(golden_dir, "sb_density"))
    assert np.max(np.abs(dens - gold)) < 1e-12
    assert np.array_equal(dens, gold)
    p = D.shockbubble_problem()
    st = D.run(p, coracle, 0.2, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps'] == 170
    assert claw.solver.status['cflmax'] == st['cflmax']
    assert claw.solver.status['dtmin'] == st['dtmin']
    assert np.array_equal(claw.frames[1].state.q, p.q)


def test_shockbubble_device_callbacks(golden_dir):
    """Same run with the built-in device BC + device source: identical arithmetic => identical bits."""
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw, device_callbacks=True)
    dens = claw.frames[claw.nout].state.q[0, :, :]
    gold = np.loadtxt(os.path.join(golden_dir, "sb_density"))
    assert np.array_equal(dens, gold)


def test_acoustics2d_golden(coracle, golden_dir):
    """test/test_examples.py:239-254: 100x100 dim-split MC, pressure vs test/acoustics2D_solution.
    Reference gate: 2-norm < 1e-14 with ITS numpy/libm (the initial condition holds a cos());
    here the IC is built by this container's numpy, which leaves 1-ulp differences in the IC
    (flang-built reference gives 1.02e-14 here too).  Gate: 2e-14, and bit equality
    with the oracle replay from the same IC."""
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw)
    pres = claw.frames[claw.nout].state.q[0, :, :]
    gold = np.loadtxt(os.path.join(golden_dir, "acoustics2D_solution"))
    assert np.linalg.norm(pres - gold) < 2e-14
    p = D.acoustics2d_problem()
    D.run(p, coracle, 0.12, 10)
    assert np.array_equal(claw.frames[claw.nout].state.q, p.q)


def test_acoustics1d_scalar(coracle):
    """test/test_examples.py:59-67: one-period L1 error 0.00104856594174 (gate 1e-5)."""
    import pyclaw_amd as pyclaw
    err, claw = problems.acoustics1D(pyclaw)
    assert abs(err - 0.00104856594174) < 1e-5
    assert abs(err - 0.00104856594174) < 1e-13
    p = D.acoustics1d_problem()
    D.run(p, coracle, 1.0, 5)
    assert np.array_equal(claw.frames[5].state.q, p.q)


def test_advection1d_c1(coracle):
    """BASELINE config[0]: 1-D advection, 1000 cells, classic -- vs the oracle replay."""
    import pyclaw_amd as pyclaw
    claw = problems.advection1D(pyclaw)
    p = D.advection1d_problem(1000)
    D.run(p, coracle, 1.0, 10)
    assert np.array_equal(claw.frames[10].state.q, p.q)


# ---- PCL_MATH_FAST: FMA + reciprocal-multiply division.  Not bit-identical; the north-star
# ---- tolerance is rtol = 1e-12 against the reference.
RTOL = 1e-12


def test_shockbubble_golden_fast(golden_dir):
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw, math='fast')
    dens = claw.frames[claw.nout].state.q[0, :, :]
    gold = np.loadtxt(os.path.join(golden_dir, "sb_density"))
    assert claw.solver.status['numsteps'] == 170
    assert np.max(np.abs(dens - gold)) < 1e-12                 # the reference's own gate
    assert np.max(np.abs(dens - gold) / np.abs(gold)) < RTOL


def test_acoustics2d_fast(coracle):
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw, math='fast')
    p = D.acoustics2d_problem()
    D.run(p, coracle, 0.12, 10)
    q = claw.frames[claw.nout].state.q
    scale = np.abs(p.q).max()
    assert np.max(np.abs(q - p.q)) < RTOL * scale


def test_shockbubble_unsplit_app_config(coracle):
    """apps/euler/2d/shockbubble/shockbubble.py:180-181 runs the UNSPLIT algorithm with
    order_trans=2 (no golden in the reference for it): replay against the oracle driver."""
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw, mx=80, my=20, tfinal=0.05, dim_split=False, order_trans=2)
    p = D.shockbubble_problem(mx=80, my=20, dim_split=False, order_trans=2)
    st = D.run(p, coracle, 0.05, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps']
    assert np.array_equal(claw.frames[1].state.q, p.q)


def test_acoustics2d_unsplit_reflecting(coracle):
    """apps/acoustics/2d/homogeneous/acoustics.py:21,31-53: unsplit, reflecting/outflow walls."""
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw, mx=60, my=50, tfinal=0.1, nout=2, dim_split=0, run=False)
    s = claw.solver
    s.order_trans = 2
    s.bc_lower[0] = pyclaw.BC.reflecting
    s.bc_lower[1] = pyclaw.BC.reflecting
    claw.run()
    p = D.acoustics2d_problem(mx=60, my=50, dim_split=False, order_trans=2,
                              bcs=([D.REFLECTING, D.REFLECTING], [D.OUTFLOW, D.OUTFLOW]))
    D.run(p, coracle, 0.1, 2)
    assert np.array_equal(claw.frames[2].state.q, p.q)


def test_strang_source_and_start_step(coracle):
    """src_split=2 (Strang) + a start_step hook: exercises the copied device backup path
    (solver.py:660,690) through a rejected step."""
    import pyclaw_amd as pyclaw
    calls = []
    claw = problems.shockbubble(pyclaw, mx=40, my=10, tfinal=0.02, run=False, dt_initial=0.05)
    claw.solver.src_split = 2
    claw.solver.start_step = lambda solver, solution: calls.append(solution.t)
    claw.run()
    p = D.shockbubble_problem(mx=40, my=10, dt_initial=0.05)
    p.src_split = 2
    st = D.run(p, coracle, 0.02, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps'] and p.nrejected >= 1
    assert len(calls) == st['numsteps'] + p.nrejected
    assert np.array_equal(claw.frames[1].state.q, p.q)


def test_controller_ascii_frames_and_gauges(tmp_path):
    """Controller.output_format='ascii' + gauges (controller.py:225-300, solver.py:731-741):
    frames read back equal the kept copies to the file's 9 digits; the gauge series (device gather,
    pcl_get_cells) has one line per accepted step and reproduces the frame values at output times bit for bit."""
    import pyclaw_amd as pyclaw
    claw = problems.acoustics2D(pyclaw, mx=60, my=50, nout=3, run=False)
    grid = claw.solution.state.grid
    grid.gauge_path = str(tmp_path / "_gauges") + os.sep
    grid.add_gauges([(0.25, 0.5), (0.9, 0.1)])
    assert grid.gauges == [[7, 12], [27, 2]]            # floor(x/d): the reference's index rule (grid.py:537)
    claw.output_format = 'ascii'
    claw.outdir = str(tmp_path)
    claw.run()
    for k, kept in enumerate(claw.frames):
        back = pyclaw.Solution(k, path=str(tmp_path))
        assert back.t == float("%18.8e" % kept.t)
        assert np.allclose(back.state.q, kept.state.q, rtol=1e-8, atol=1e-99)
        assert back.state.grid.n == [60, 50] and back.state.grid.lower == [-1.0, -1.0]
    for g, name in zip(grid.gauges, ("gauge0.25_0.5.txt", "gauge0.9_0.1.txt")):
        rows = [[float(v) for v in line.split()] for line in open(os.path.join(grid.gauge_path, name))]
        times = [r[0] for r in rows]
        assert times[0] == 0.0 and all(b > a for a, b in zip(times, times[1:]))
        assert len(rows) == claw.nout * claw.solver.status['numsteps'] + 1     # status counts the last evolve call
        series = {r[0]: r[1:] for r in rows}
        for kept in claw.frames:
            assert series[kept.t] == list(kept.state.q[:, g[0], g[1]])


def test_restart_from_block_checkpoint(tmp_path):
    """frames written as block checkpoints are restart points: a run restarted from frame 2
    (Solution(frame, format='block') + Controller.start_frame) ends bit-equal to the uninterrupted run
    (fixed dt, so both runs take the same steps)."""
    import pyclaw_amd as pyclaw

    def make(nout, tfinal):
        claw = problems.acoustics2D(pyclaw, mx=50, my=40, nout=nout, tfinal=tfinal, run=False)
        claw.solver.dt_variable = False
        claw.solver.dt_initial = 0.003
        claw.output_format = 'block'
        claw.outdir = str(tmp_path)
        return claw

    full = make(4, 0.12)
    full.run()
    again = make(2, 0.12)
    again.solution = pyclaw.Solution(2, path=str(tmp_path), format='block')
    assert again.solution.t == full.frames[2].t
    assert again.solution.state.aux_global['cc'] == 2.0          # problem scalars travel in the header
    again.start_frame = 2
    again.outdir = str(tmp_path / "restart")
    again.run()
    assert again.frames[-1].t == full.frames[4].t
    assert np.array_equal(again.frames[-1].state.q, full.frames[4].state.q)
    assert os.path.exists(os.path.join(again.outdir, "claw.ckpt0004.json"))


@pytest.mark.parametrize("device_callbacks", [False, True])
def test_shockbubble_multi_tile_replay(coracle, device_callbacks):
    """The shock-bubble app on a 520 x 150 grid (several tiles in both directions, ragged edges) to t = 0.03:
    source term, inflow/reflecting/outflow BCs, adaptive dt with its CFL feedback, the jump-free and
    absent-family shortcuts -- every accepted/rejected step and the final state equal the oracle driver's
    replay bit for bit (the reference's own golden, 160 x 40, is smaller than one x tile)."""
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw, mx=520, my=150, tfinal=0.03, device_callbacks=device_callbacks,
                                dt_initial=0.005 * 160.0 / 520.0)
    p = D.shockbubble_problem(mx=520, my=150)
    p.dt_initial = 0.005 * 160.0 / 520.0
    st = D.run(p, coracle, 0.03, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps'] and st['numsteps'] > 40
    assert claw.solver.status['cflmax'] == st['cflmax'] and claw.solver.status['dtmin'] == st['dtmin']
    assert np.array_equal(claw.frames[1].state.q, p.q)


def test_fast_mode_other_kernel_families(coracle):
    """The second arithmetic mode (FMA contraction + reciprocal-multiply division) through the kernels the
    golden tests above do not reach: unsplit Euler (transverse solves), 3-D acoustics sweeps, SharpClaw Euler
    (WENO5 + two Riemann solves, the occupancy-4 build with its spills) -- each within rtol 1e-12 of the
    oracle after a short run (same step count: the dt history may differ in the last bits)."""
    import pyclaw_amd as pyclaw
    # unsplit Euler, 30 steps' worth
    claw = problems.shockbubble(pyclaw, mx=200, my=60, tfinal=0.02, dim_split=False, order_trans=2, math='fast',
                                dt_initial=0.004)
    p = D.shockbubble_problem(mx=200, my=60, dim_split=False, order_trans=2)
    p.dt_initial = 0.004
    st = D.run(p, coracle, 0.02, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps']
    q = claw.frames[1].state.q
    # (the y momentum of this early-time state is rounding residue, ~1e-20: scale with the state, not per component)
    assert np.max(np.abs(q - p.q)) < RTOL * np.abs(p.q).max()
    for m in (0, 1, 3):
        assert np.max(np.abs(q[m] - p.q[m])) < RTOL * np.abs(p.q[m]).max(), m
    # 3-D acoustics, dim-split
    claw = problems.acoustics3D(pyclaw, mx=40, my=12, mz=10, tfinal=0.2, nout=1, math='fast')
    p = D.acoustics3d_problem('hom', mx=40, my=12, mz=10)
    st = D.run(p, coracle, 0.2, 1)[-1]
    assert claw.solver.status['numsteps'] == st['numsteps']
    assert np.max(np.abs(claw.frames[1].state.q - p.q)) < RTOL * np.abs(p.q).max()
    # SharpClaw Euler: one right-hand side through the C ABI
    import ctypes as C
    from pyclaw_amd import _lib as L
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    mx, my = 90, 70
    q0 = np.empty((5, mx + 6, my + 6), order="F")
    rho = 0.5 + rng.random(q0.shape[1:]); u = rng.random(q0.shape[1:]) - 0.5; v = rng.random(q0.shape[1:]) - 0.5
    q0[0], q0[1], q0[2] = rho, rho * u, rho * v
    q0[3] = (0.5 + rng.random(q0.shape[1:])) / 0.4 + 0.5 * rho * (u * u + v * v)
    q0[4] = rng.random(q0.shape[1:])
    ref, cfl_ref = coracle.sharp_flux2(O.RP_EULER5_2D, [1.4, 0.4], 2, 5, 0, 3, mx, my, q0, None, 0.01, 0.012, 1e-3)
    cfg = L.Config()
    cfg.ndim = 2; cfg.n[0], cfg.n[1] = mx, my; cfg.mbc = 3; cfg.meqn = cfg.mwaves = 5; cfg.rp = 11
    cfg.method[1] = 2; cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.d[0], cfg.d[1] = 0.01, 0.012; cfg.kind = 1; cfg.lim_type = 2; cfg.math = 1
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    try:
        L.check(L.lib().pcl_put_q(h, L.d(q0), 1))
        cfl = C.c_double()
        L.check(L.lib().pcl_sharp_dq(h, 1e-3, C.cast(C.byref(cfl), L.dp)))
        L.check(L.lib().pcl_select(h, 3))
        dq = np.zeros_like(q0)
        L.check(L.lib().pcl_get_q(h, L.d(dq), 1))
    finally:
        L.lib().pcl_destroy(h)
    inner = (slice(None), slice(3, -3), slice(3, -3))
    assert abs(cfl.value - cfl_ref) < 1e-12 * cfl_ref
    assert np.max(np.abs(dq[inner] - ref[inner])) < 1e-11 * np.abs(ref[inner]).max()


@pytest.mark.parametrize("dim_split", [True, False])
def test_fused_source_term_equals_separate_kernel(coracle, monkeypatch, dim_split):
    """The Godunov-split radial source applied inside the y pass / y phase (pcl_fuse_source) == the separate source
    kernel (PCL_FUSE_SRC=0) == the oracle replay, bit for bit, with a rejected step on the way (dt_initial too large)."""
    import pyclaw_amd as pyclaw
    res = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("PCL_FUSE_SRC", fuse)
        claw = problems.shockbubble(pyclaw, tfinal=0.05, device_callbacks=True, dt_initial=0.02, run=False,
                                    dim_split=dim_split)
        claw.run()
        assert claw.solver._src_fused == (fuse == "1")
        res[fuse] = (claw.frames[claw.nout].state.q.copy(), dict(claw.solver.status))
    assert np.array_equal(res["1"][0], res["0"][0]) and res["1"][1] == res["0"][1]
    p = D.shockbubble_problem(dt_initial=0.02, dim_split=dim_split)
    st = D.run(p, coracle, 0.05, 1)[-1]
    assert p.nrejected >= 1 and res["1"][1]["numsteps"] == st["numsteps"]
    assert np.array_equal(res["1"][0], p.q)


def test_fused_source_when_cfl_equals_cfl_max(monkeypatch):
    """clawpack.py:153 returns without the source term when cfl >= cfl_max, solver.py:668 accepts cfl <= cfl_max: at
    equality the reference keeps the hyperbolic step WITHOUT its source.  With the source fused into the y pass that
    step is redone unfused: same result as the separate-kernel path."""
    import pyclaw_amd as pyclaw
    out = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("PCL_FUSE_SRC", fuse)
        probe = problems.shockbubble(pyclaw, tfinal=1.0, device_callbacks=True, run=False)
        probe.solver.setup(probe.solution)
        probe.solver.dt = probe.solver.dt_initial
        probe.solver.evolve_to_time(probe.solution)              # one step: its Courant number
        c1 = probe.solver.cfl.get_cached_max()
        probe.solver.teardown()
        claw = problems.shockbubble(pyclaw, tfinal=1.0, device_callbacks=True, run=False)
        claw.solver.cfl_max = c1                                  # the first step lands exactly on cfl_max
        claw.solver.cfl_desired = 0.9 * c1
        claw.solver.setup(claw.solution)
        claw.solver.dt = claw.solver.dt_initial
        for _ in range(3):
            claw.solver.evolve_to_time(claw.solution)
        claw.solver._pull(claw.solution.state) if hasattr(claw.solver, "_pull") else None
        out[fuse] = (claw.solution.state.q.copy(), claw.solution.t, c1)
        claw.solver.teardown()
    assert out["1"][2] == out["0"][2] and out["1"][1] == out["0"][1]
    assert np.array_equal(out["1"][0], out["0"][0])


@pytest.mark.parametrize("ti", ["SSP104", "SSP33", "Euler"])
def test_sharpclaw_shockbubble_dq_src(coracle, ti):
    """apps/euler/2d/shockbubble/shockbubble.py:173-176: SharpClawSolver2D (WENO5) with dq_src = dq_Euler_radial.  The
    device twin evaluated inside the last pass of every stage (pcl_sharp_fuse_dq_src) == the numpy callback through
    the host == the oracle driver's replay, bit for bit, for every RK combination the pass can end in."""
    import pyclaw_amd as pyclaw
    cfl = {} if ti == "SSP104" else dict(cfl_max=0.5, cfl_desired=0.45)    # the low-stage schemes need a smaller step
    res = {}
    for dev in (True, False):
        claw = problems.shockbubble(pyclaw, mx=96, my=40, tfinal=0.1, device_callbacks=dev, solver_type='sharpclaw',
                                    time_integrator=ti, dt_initial=0.002, run=False)
        for k, v in cfl.items():
            setattr(claw.solver, k, v)
        claw.run()
        assert claw.solver._dq_src_fused == dev
        res[dev] = (claw.frames[claw.nout].state.q.copy(), dict(claw.solver.status))
    assert res[True][1] == res[False][1]
    assert np.array_equal(res[True][0], res[False][0])
    p = D.shockbubble_problem(mx=96, my=40, solver_type='sharpclaw', time_integrator=ti, dt_initial=0.002, **cfl)
    st = D.run(p, coracle, 0.1, 1)[-1]
    assert res[True][1]["numsteps"] == st["numsteps"] and res[True][1]["numsteps"] > 3
    assert np.array_equal(res[True][0], p.q)
    # and the source is not a no-op
    q_nosrc = problems.shockbubble(pyclaw, mx=96, my=40, tfinal=0.1, device_callbacks=True, solver_type='sharpclaw',
                                   time_integrator=ti, dt_initial=0.002, with_src=False, run=False)
    for k, v in cfl.items():
        setattr(q_nosrc.solver, k, v)
    q_nosrc.run()
    assert np.abs(q_nosrc.frames[1].state.q - res[True][0]).max() > 1e-5


def test_device_dq_source_falls_back_to_host_when_not_fusable():
    """EulerRadialDqSource on a configuration the fused kernel is not built for (lim_type 1, the tvd2 reconstruction) runs as the numpy
    callable it also is; pcl_sharp_fuse_dq_src itself refuses that configuration."""
    import ctypes
    import pyclaw_amd as pyclaw
    from pyclaw_amd import _lib
    claw = problems.shockbubble(pyclaw, mx=64, my=32, tfinal=0.01, device_callbacks=True, solver_type='sharpclaw',
                                dt_initial=0.002, run=False)
    claw.solver.lim_type = 1
    ref = problems.shockbubble(pyclaw, mx=64, my=32, tfinal=0.01, device_callbacks=False, solver_type='sharpclaw',
                               dt_initial=0.002, run=False)
    ref.solver.lim_type = 1
    claw.run()
    ref.run()
    assert not claw.solver._dq_src_fused
    assert np.isfinite(ref.frames[1].state.q).all()
    assert np.array_equal(claw.frames[1].state.q, ref.frames[1].state.q)
    src = claw.solver.dq_src
    claw2 = problems.shockbubble(pyclaw, mx=64, my=32, tfinal=0.01, device_callbacks=True, solver_type='sharpclaw',
                                 run=False)
    claw2.solver.lim_type = 1
    claw2.solver.setup(claw2.solution)
    rc = _lib.lib().pcl_sharp_fuse_dq_src(claw2.solver._h, 1, _lib.d(src.params), 2)
    assert rc != 0
    claw2.solver.teardown()


def test_shockbubble_golden_strict_math(golden_dir):
    """math='strict' (PCL_MATH_STRICT, the third build of the kernels: exact + IEEE quotients for underflow-range
    numerators) gives the golden density bit for bit as well, classic and SharpClaw stage kernels both load."""
    import pyclaw_amd as pyclaw
    claw = problems.shockbubble(pyclaw, math='strict')
    gold = np.loadtxt(os.path.join(golden_dir, "sb_density"))
    assert np.array_equal(claw.frames[claw.nout].state.q[0, :, :], gold)
    a = problems.acoustics2D(pyclaw, mx=40, my=40, solver_type='sharpclaw', tfinal=0.05, nout=1, run=False)
    b = problems.acoustics2D(pyclaw, mx=40, my=40, solver_type='sharpclaw', tfinal=0.05, nout=1, run=False)
    b.solver.math = 'strict'
    a.run()
    b.run()
    assert np.array_equal(a.frames[1].state.q, b.frames[1].state.q)
