#!/usr/bin/env python
"""
bench.py -- headline benchmark: Mcell.steps/s of the classic dimension-split step on the
2-D Euler shock-bubble problem (BASELINE.json configs[2]: 4096 x 4096 cells on one MI355X).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nx 4096 --ny 4096]

One "step" = one accepted pass of the hot path on the block resident in HBM: ghost-cell
fill (+ halo exchange when N > 1), x sweep, y sweep, device CFL reduction, the 8-byte CFL
read-back and the host's accept/retake + dt update -- i.e. solver.evolve_to_time(solution)
taking one step, exactly what the reference's Controller drives.  The source term is off
(the "classic step" figure of SURVEY 8d); inputs are synthetic (the shock-bubble initial
condition evaluated on the benchmark grid).

N > 1 (launched by torch.distributed.run, one rank per GPU): weak scaling by default -- every rank owns
a 4096 x 4096 block of a (4096*px) x (4096*py) grid, halo exchange over RCCL, CFL all-reduce.
`--global NX NY` fixes the GLOBAL grid instead (strong scaling): `--gpus 8 --global 8192 8192` is BASELINE
configs[3] (8192 x 8192 on a 2 x 4 processor grid, 4096 x 2048 cells per GPU); config.workload says which.

Besides the headline (the shock-bubble initial condition of BASELINE configs[2]) the default single-GPU line
carries the same K steps on two other states of the same grid, because the kernels' cost depends on the state:
`dense_state` (a jump at every interface, all five wave families present everywhere: the input-independent
figure for the kernels) and `developed_state` (the reference regression's own t = 0.2 solution, shock through
the bubble, interpolated to the benchmark grid).  When K is small a `sustained` object adds a 1000-step run.

Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
"roofline" (dominant sweep kernel, HIP-event timed inside the timed region) and
"cpu_baseline" (the reference Fortran / the C port of it timed on this box's host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
BYTES_PER_CELL_SWEEP = 2 * 5 * 8   # read + write of q (5 doubles) per directional pass


def build(nx_global, ny_global, math, unsplit=False, with_src=False):
    import pyclaw_amd as pyclaw
    from apps import problems
    # dt_initial scaled with dx like the reference test (0.005 at dx=1/80) -> first CFL ~ 0.4-0.5
    dt0 = 0.005 * (2.0 / nx_global) / (2.0 / 160.0)
    claw = problems.shockbubble(pyclaw, mx=nx_global, my=ny_global, device_callbacks=True,
                                with_src=with_src, dt_initial=dt0, run=False, math=math,
                                dim_split=not unsplit, order_trans=2)
    return claw


def dense_state(claw, seed=0):
    """Overwrite the initial condition with the smooth random Euler state of SURVEY 8d: waves of every
    family at every interface, nothing for the kernels' jump-free / absent-family shortcuts to skip."""
    rng = np.random.default_rng(seed)
    q = claw.solution.state.q
    shape = q.shape[1:]
    q[0] = 1.0 + 0.1 * rng.random(shape)
    q[1] = 0.1 * rng.random(shape)
    q[2] = 0.05 * rng.random(shape)
    q[3] = 2.5 + 0.1 * rng.random(shape)
    q[4] = rng.random(shape)
    return claw


def developed_state(claw):
    """Overwrite the initial condition with a developed flow: the reference regression
    (test/euler/2d/shockbubble.py: 160 x 40 cells, t = 0.2, source term on -- the run behind the golden
    test/sb_density) computed here on the GPU, then interpolated bilinearly to the benchmark grid.  The
    interpolation is written a + w*(b - a), so regions the coarse solution leaves exactly constant (gas the
    shock has not reached) stay exactly constant, as they would in a run on the fine grid."""
    import pyclaw_amd as pyclaw
    from apps import problems
    coarse = problems.shockbubble(pyclaw, device_callbacks=True).frames[-1].state.q      # (5, 160, 40)
    q = claw.solution.state.q
    nx, ny = q.shape[1:]
    cx, cy = coarse.shape[1:]

    def stencil(n, c):
        f = (np.arange(n) + 0.5) * (float(c) / n) - 0.5
        f = np.clip(f, 0.0, c - 1.0)
        i0 = np.minimum(np.floor(f).astype(np.int64), c - 2)
        return i0, f - i0
    ix, wx = stencil(nx, cx)
    iy, wy = stencil(ny, cy)
    for m in range(q.shape[0]):
        a = coarse[m]
        rows = a[ix, :] + wx[:, None] * (a[ix + 1, :] - a[ix, :])                     # (nx, cy)
        q[m] = rows[:, iy] + wy[None, :] * (rows[:, iy + 1] - rows[:, iy])
    return claw


def build_sphere(nx, ny, math, solver_type):
    """apps/shallow-sphere/shallow_4_Rossby_Haurwitz_wave.py on an nx x ny computational grid (aux and q from the
    app's own setaux / qinit formulas, apps/shallow_sphere.py)."""
    import pyclaw_amd as pyclaw
    from apps import shallow_sphere as ss
    claw = ss.shallow_sphere(pyclaw, mx=nx, my=ny, run=False, math=math, solver_type=solver_type)
    claw.solver.dt_initial = 0.1 * 40.0 / nx      # the app's 0.1 at 40x20
    return claw


def build_sharp(nx, ny, math, dq_src=False):
    """SharpClaw (WENO5 + SSP104, 10 right-hand sides per step) on the shock-bubble problem; dq_src: the app's
    dq_Euler_radial (apps/euler/2d/shockbubble/shockbubble.py:173-176) as its device twin, fused in the last pass."""
    import pyclaw_amd as pyclaw
    from apps import problems
    x = pyclaw.Dimension('x', 0.0, 2.0, nx)
    y = pyclaw.Dimension('y', 0.0, 0.5, ny)
    state = pyclaw.State(pyclaw.Grid([x, y]), 5, 1)
    state.aux_global['gamma'] = problems.gamma
    state.aux_global['gamma1'] = problems.gamma1
    problems.sb_qinit(state)
    problems.sb_auxinit(state)
    solver = pyclaw.SharpClawSolver2D()
    solver.rp = pyclaw.riemann.rp_euler_5wave_2d
    solver.mwaves = 5
    solver.lim_type = 2
    solver.math = math
    rinf, vinf, einf = problems.shock_state()
    solver.user_bc_lower = pyclaw.ConstantStateBC([rinf, rinf * vinf, 0., einf, 0.])
    solver.bc_lower = [pyclaw.BC.custom, pyclaw.BC.reflecting]
    solver.bc_upper = [pyclaw.BC.outflow, pyclaw.BC.outflow]
    solver.aux_bc_lower = [pyclaw.BC.outflow] * 2
    solver.aux_bc_upper = [pyclaw.BC.outflow] * 2
    solver.dt_initial = 0.4 * (2.0 / nx)
    if dq_src:
        solver.dq_src = pyclaw.EulerRadialDqSource(problems.gamma1, 2)
    claw = pyclaw.Controller()
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    return claw


def build3d(n, math, unsplit=False):
    """3-D synthetic workload: the reference's 3-D acoustics app (test/acoustics/3d/acoustics.py, 'hom' set-up:
    dim-split, periodic) on an n[0] x n[1] x n[2] grid with a two-material aux field."""
    import pyclaw_amd as pyclaw
    from apps import problems
    if isinstance(n, int):
        n = (n, n, n)
    claw = problems.acoustics3D(pyclaw, test='het' if unsplit else 'hom', mx=n[0], my=n[1], mz=n[2], run=False,
                                math=math)        # 'het' = the reference's unsplit step3 set-up (order_trans 22)
    st = claw.solution.state
    X, Y, Z = st.grid.c_center
    st.aux[0] = 1.0 + (X >= 0.)                     # impedance 1 | 2
    st.aux[1] = 1.0 + (X >= 0.)                     # sound speed 1 | 2
    r = np.sqrt((X + 0.5) ** 2 + Y ** 2 + Z ** 2)
    st.q[0] = (np.abs(r - 0.3) <= 0.1) * (1. + np.cos(np.pi * (r - 0.3) / 0.1))
    claw.solver.dt_initial = 0.4 * (2.0 / max(n)) / 2.0
    return claw


PMC_FILE = "r03_pmc_hbm.json"     # written by tools/profile_r03.sh + tools/pmc_summary.py


def pmc_traffic(mode, needles, default_grid):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/<round>_pmc_hbm.json:
    FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes, the same bench command).  PMC counters cannot be
    read from inside this process, so the number is only reported for the configuration it was collected on
    (`default_grid` says whether this run is that one), and `traffic_source` names the committed profile and the commit
    of the kernel build it belongs to.  mode: key of the profile's "modes"; needles: substrings the kernel name must
    contain.  Returns (bytes or None, source string or None)."""
    if not default_grid:
        return None, None
    try:
        with open(os.path.join(ROOT, "profiles", PMC_FILE)) as f:
            doc = json.load(f)
        for name, v in doc["modes"][mode].items():
            if all(n in name for n in needles):
                return v["hbm_bytes_per_launch_corrected"], (
                    "from committed profile profiles/%s (collected at commit %s)" % (PMC_FILE, doc.get("commit", "?")))
    except Exception:
        pass
    return None, None


def copy_ceiling():
    """What a pure 5-planes-in / 5-planes-out copy of the sweep's tile shape reaches on an MI355X (nontemporal accesses),
    from the committed run of tools/ubench/copy_rates.hip: the practical ceiling next to the 8 TB/s data-sheet peak.
    Returns (GB/s or None, source)."""
    try:
        best = 0.0
        with open(os.path.join(ROOT, "profiles", "r02_copy_rates.txt")) as f:
            for line in f:
                if line.startswith("tile") and "GB/s" in line:
                    best = max(best, float(line.split()[-2]))
        return (best or None), "profiles/r02_copy_rates.txt (tools/ubench/copy_rates.hip, best tile-shaped copy)"
    except Exception:
        return None, None


def cpu_baseline(nx, ny, max_seconds=30.0):
    """Time the reference's CPU path on ONE core on a bounded sample of the same workload:
    whole dim-split steps (x then y sweep) of the same Euler state on an nx x (rows) slab.
    kind 'reference' = oracle/_ref (the reference Fortran itself, flang-built); 'port' = the
    C restatement.  The slab height is chosen so the sample takes ~10-20 s."""
    from oracle import oracle as O
    from oracle import driver as D
    if O.RefEuler2D.available():
        be, kind = O.RefEuler2D(), "reference"
    else:
        be, kind = O.COracle(), "port"
    # ~1.3 Mcell.steps/s/core was measured at survey time: 16 M cells ~ 13 s
    sy = max(8, min(ny, int(16.0e6 / nx)))
    p = D.shockbubble_problem(mx=nx, my=sy, with_src=False)
    p.d = (2.0 / nx, 0.5 / ny)
    p.dt_initial = 0.005 * (2.0 / nx) / (2.0 / 160.0)
    D.setup(p)
    p.dt = p.dt_initial
    t0 = time.perf_counter()
    nsteps = 0
    while True:
        D.step_hyperbolic(p, be)
        nsteps += 1
        el = time.perf_counter() - t0
        if el > 10.0 or nsteps >= 3 or el * (nsteps + 1) / nsteps > max_seconds:
            break
    el = time.perf_counter() - t0
    return {"value": nx * sy * nsteps / el / 1e6, "unit": "Mcell*steps/s", "cores": 1, "kind": kind,
            "sample": "%d dim-split steps of the same shock-bubble state on a %dx%d slab (%.1f s)"
                      % (nsteps, nx, sy, el)}


def _cpu_worker(args):
    """one process of the all-cores leg: dim-split steps of the shock-bubble state on its own slab"""
    nx, sy, ny, nsteps = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import oracle as O
    from oracle import driver as D
    be = O.RefEuler2D() if O.RefEuler2D.available() else O.COracle()
    p = D.shockbubble_problem(mx=nx, my=sy, with_src=False)
    p.d = (2.0 / nx, 0.5 / ny)
    p.dt_initial = 0.005 * (2.0 / nx) / (2.0 / 160.0)
    D.setup(p)
    p.dt = p.dt_initial
    t0 = time.perf_counter()
    for _ in range(nsteps):
        D.step_hyperbolic(p, be)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(nx, ny):
    """the reference is one process per core (PetClaw: one MPI rank per core, one sub-domain
    each).  C = the cores this process may use; C workers step C slabs of the same state at the same time and
    the aggregate rate is reported.  The slabs are independent (no halo exchange between them), so this is
    an upper bound for the reference's own MPI run on these cores."""
    import multiprocessing as mp
    from oracle import oracle as O
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    # a one-GPU box's CPU share is 16 cores (the affinity mask shows the whole host); 256 workers x ~1 GB also ran
    # into the host-memory cap and took 130 s.  PCL_BENCH_CPU_PROCS overrides.
    cores = max(1, min(cores, int(os.environ.get("PCL_BENCH_CPU_PROCS", "16"))))
    kind = "reference" if O.RefEuler2D.available() else "port"
    sy, nsteps = max(8, min(ny, int(8.0e6 / nx))), 2          # ~3 s per step per core at 5 Mcell/s/core
    ctx = mp.get_context("spawn")                              # no fork of a process that holds a GPU context
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        times = pool.map(_cpu_worker, [(nx, sy, ny, nsteps)] * cores)
    wall = time.perf_counter() - t0
    slowest = max(times)
    return {"value": cores * nx * sy * nsteps / slowest / 1e6, "unit": "Mcell*steps/s", "cores": cores, "kind": kind,
            "sample": "%d processes, each %d dim-split steps on its own %dx%d slab of the same state, at the same "
                      "time; slowest %.1f s (pool wall %.1f s); no halo exchange between slabs (upper bound for "
                      "one MPI rank per core)" % (cores, nsteps, nx, sy, slowest, wall)}


def timed_run(claw, steps, warmup):
    """W warmup steps, then exactly K accepted steps between barrier + device sync on both sides.
    Returns (max-over-ranks seconds, per-kernel ms sums, launch counts, result finite)."""
    import ctypes
    from pyclaw_amd import parallel, _lib
    solver, solution = claw.solver, claw.solution
    solver.setup(solution)
    solver.dt = solver.dt_initial
    L = _lib.lib()
    h = solver._h
    solver.begin_resident(solution)
    for _ in range(warmup):
        solver.evolve_to_time(solution)
    # HIP events around the sweep launches of every N-th step of the timed region.  Every step (N = 1) for short
    # runs; every 8th for long ones: the event records sit between the kernels, and with one pair around every launch
    # the x pass itself runs 4 % slower (0.265 vs 0.253 ms) and the step 2-3 % -- the average is over >= 125 launches
    # spread uniformly over the region either way (counts: config.launches_timed).
    every = int(os.environ.get("PCL_BENCH_TIMING", "0")) or (8 if steps >= 200 else 1)
    _lib.check(L.pcl_kernel_timing(h, every))
    n0 = ctypes.c_long()
    _lib.check(L.pcl_step_count(h, ctypes.byref(n0)))
    parallel.barrier()
    _lib.check(L.pcl_sync(h))
    t0 = time.perf_counter()
    for _ in range(steps):
        solver.evolve_to_time(solution)
    _lib.check(L.pcl_sync(h))
    parallel.barrier()
    t1 = time.perf_counter()
    elapsed = parallel.allreduce_max_host(t1 - t0)
    ms = np.zeros(3)
    nl = np.zeros(3, dtype=np.int64)
    _lib.check(L.pcl_kernel_timing_read(h, _lib.d(ms), nl.ctypes.data_as(ctypes.POINTER(ctypes.c_long))))
    # slot 2: the one-kernel form of the dim-split 2-D step (classic_fused.hpp) + how many steps ran in each form
    ms2, n2, s1, s0 = ctypes.c_double(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    _lib.check(L.pcl_step_form_stats(h, ctypes.byref(ms2), ctypes.byref(n2), ctypes.byref(s1), ctypes.byref(s0)))
    ms[2], nl[2] = ms2.value, n2.value
    timed_run.forms = {"one_kernel_steps": int(s1.value), "two_pass_steps": int(s0.value)}
    _lib.check(L.pcl_kernel_timing(h, 0))
    n1 = ctypes.c_long()
    _lib.check(L.pcl_step_count(h, ctypes.byref(n1)))
    timed_run.attempted = n1.value - n0.value      # steps (or SharpClaw right-hand sides) incl. rejected ones
    solver.end_resident(solution)
    finite = bool(np.isfinite(solution.state.q).all())
    solver.teardown()
    return elapsed, ms, nl, finite


def app_run_object(nxg, nyg, cells_total):
    """The shock-bubble problem itself on the benchmark grid, from t = 0 to the regression's t = 0.2 (test/euler/2d/
    shockbubble.py:97-166: inflow / reflecting / outflow sides, adaptive dt at cfl_desired 0.45, radial source term
    after every step -- fused into the y pass), through solver.evolve_to_time in four quarters of the time interval:
    the whole-run rate and the rate per quarter, i.e. how the headline (the first steps, mostly undisturbed gas) decays
    as the shock crosses the bubble and the flow fills the domain."""
    from pyclaw_amd import _lib
    claw = build(nxg, nyg, "exact", False, with_src=True)
    solver, solution = claw.solver, claw.solution
    solver.setup(solution)
    solver.dt = solver.dt_initial
    solver.max_steps = 10 ** 8
    L = _lib.lib()
    solver.begin_resident(solution)
    quarters = []
    total_steps, total_s = 0, 0.0
    for k in range(1, 5):
        _lib.check(L.pcl_sync(solver._h))
        t0 = time.perf_counter()
        st = solver.evolve_to_time(solution, 0.05 * k)
        _lib.check(L.pcl_sync(solver._h))
        el = time.perf_counter() - t0
        n = int(st['numsteps'])
        quarters.append({"t_end": 0.05 * k, "steps": n, "seconds": el, "ms_per_step": el / max(n, 1) * 1e3,
                         "value": cells_total * n / el / 1e6, "cflmax": float(st['cflmax'])})
        total_steps += n
        total_s += el
    solver.end_resident(solution)
    finite = bool(np.isfinite(solution.state.q).all())
    solver.teardown()
    return {"value": cells_total * total_steps / total_s / 1e6, "unit": "Mcell*steps/s", "steps": total_steps,
            "seconds": total_s, "ms_per_step": total_s / max(total_steps, 1) * 1e3, "math": "exact",
            "roofline_frac_whole_step": 160.0 * cells_total * total_steps / total_s / 1e9 / HBM_PEAK_GBS,
            "what": "apps/euler 2D shock-bubble, t = 0 .. 0.2, adaptive dt (accepted steps counted; rejected ones cost "
                    "time too), source term on (fused into the y pass), BCs on the device; per quarter of the time interval",
            "quarters": quarters, "result_finite": finite}


def pass_view(ms, nl, bytes_pass):
    """The dim-split step's launches as the solver timed them: two passes (decomposed blocks, capa / aux solvers) or
    ONE kernel that does both sweeps (step2ds_kernel, classic_fused.hpp: a single block of an aux-free solver).
    Returns (avg_ms labels, duration of the dominant launch, its algorithmic bytes).  Either launch reads q once and
    writes it once (80 B per cell for Euler): for a directional pass that is SURVEY 8(d)'s count, for the one-kernel step
    it is the minimum of THAT form (it does both passes' arithmetic on those bytes) -- the form-independent figure is
    the whole-step one, SURVEY 8(d)'s 160 B per cell and step over the wall time of a step."""
    avg = [ms[k] / max(1, nl[k]) for k in range(3)]
    lab = {}
    if nl[2] > 0:
        lab["x + y sweeps, one kernel"] = avg[2]
    if nl[0] > 0 or nl[1] > 0:
        lab["x pass"], lab["y pass"] = avg[0], avg[1]
    if nl[2] >= nl[0]:          # the form most of the region ran in (the solver picks the faster one, re-measured
        return lab, avg[2], bytes_pass              # every 256 steps: pcl_step_form_stats)
    return lab, max(avg[:2]), bytes_pass


def state_object(tag, claw, steps, warmup, cells_total, bytes_launch, describe):
    """K steps of the same solver on another state of the same grid: value + the dominant launch's roofline fraction"""
    el, ms, nl, fin = timed_run(claw, steps, warmup)
    lab, dur, byt = pass_view(ms, nl, bytes_launch)
    return {"value": cells_total * steps / el / 1e6, "unit": "Mcell*steps/s", "steps": steps,
            "ms_per_step": el / steps * 1e3, "math": "exact", "state": describe,
            "roofline_frac": byt / (dur * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "achieved_GBs": byt / (dur * 1e-3) / 1e9,
            "roofline_frac_whole_step": 2.0 * bytes_launch / (el / steps) / 1e9 / HBM_PEAK_GBS,
            "dominant_launch": "one kernel (x + y sweeps)" if nl[2] >= nl[0] else "a directional pass",
            "avg_ms": lab, "result_finite": fin}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000,
                    help="timed steps (default 1000: a timed region of >= 0.5 s at 4096^2)")
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--nx", type=int, default=4096, help="cells per GPU block in x (weak scaling)")
    ap.add_argument("--ny", type=int, default=4096, help="cells per GPU block in y (weak scaling)")
    ap.add_argument("--global", dest="glob", type=int, nargs=2, metavar=("NX", "NY"), default=None,
                    help="fix the GLOBAL grid (strong scaling): --gpus 8 --global 8192 8192 = BASELINE configs[3]")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-states", action="store_true",
                    help="skip the dense_state / developed_state / sustained objects of the default line")
    ap.add_argument("--no-app-run", dest="no_app_run", action="store_true",
                    help="skip the app_run object (the shock-bubble problem from t = 0 to 0.2 on the benchmark grid, ~10 s)")
    ap.add_argument("--state", choices=["bubble", "dense", "developed"], default="bubble",
                    help="state the main timed run starts from (profile runs of the dense / developed state: "
                         "tools/profile_round.sh); anything but bubble implies --no-states")
    ap.add_argument("--math", choices=["exact", "fast", "strict"], default="exact")
    ap.add_argument("--unsplit", action="store_true", help="unsplit algorithm with order_trans=2 (not the headline)")
    ap.add_argument("--extras", action="store_true",
                    help="also time the fast arithmetic mode (fast_math object)")
    ap.add_argument("--solver", choices=["classic", "sharpclaw"], default="classic",
                    help="sharpclaw: WENO5 + SSP104 on the same problem (single GPU; not the headline)")
    ap.add_argument("--app", choices=["bubble", "sphere"], default="bubble",
                    help="sphere: apps/shallow-sphere (Rossby-Haurwitz wave, 16 aux planes, capacity function) on an "
                         "nx x nx/2 grid, default 2048x1024 = BASELINE configs[4]'s grid; --solver classic is the "
                         "reference app (unsplit step2qcor + Coriolis source), --solver sharpclaw the configs[4] variant")
    ap.add_argument("--dq-src", dest="dq_src", action="store_true",
                    help="--solver sharpclaw: with the app's radial source (dq_src) evaluated inside the last pass")
    ap.add_argument("--ndim", type=int, default=2, choices=[2, 3],
                    help="3: 3-D dim-split acoustics on an nx^3 grid (single GPU; not the headline)")
    args = ap.parse_args()

    from pyclaw_amd import parallel, _lib
    # a multi-GPU run whose RCCL set-up fails on this node still produces a (slow, clearly labelled) line over the
    # host-staged halo wire instead of no line at all: config.halo_transport says which wire ran
    os.environ.setdefault("PCL_HALO_FALLBACK", "host")
    parallel.init()
    rank, size = parallel.rank(), parallel.world_size()
    if size != args.gpus:
        if rank == 0:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)\n"
                             % (args.gpus, size))
        sys.exit(2)

    scaling = "weak"
    if args.ndim == 3:
        # weak scaling: n^3 cells per GPU, blocks cut in y and z (x-rows stay whole)
        if args.nx == 4096:
            args.nx = 512
        args.ny = args.nx
        pd = parallel.proc_grid([args.nx, args.nx], size) if size > 1 else [1, 1]
        dims, nxg, nyg = [1] + pd, args.nx, args.nx * pd[0]
        claw = build3d((args.nx, args.nx * pd[0], args.nx * pd[1]), args.math, args.unsplit)
    elif args.app == "sphere":
        # BASELINE configs[4]: the GLOBAL grid is fixed (2048 x 1024) and cut in y only, 1 x N blocks -- the pole
        # boundary reverses whole rows (strong scaling)
        if args.nx == 4096:
            args.nx = 2048
        args.ny = args.nx // 2
        scaling = "strong" if size > 1 else "weak"
        os.environ["PCL_PROC_GRID"] = "1x%d" % size
        dims, nxg, nyg = [1, size], args.nx, args.ny
        claw = build_sphere(nxg, nyg, args.math, args.solver)
        args.nx, args.ny = claw.solution.state.q.shape[1:3]          # THIS rank's block (rank 0 prints)
    elif args.solver == "sharpclaw":
        # weak scaling like the classic line: nx x ny cells per GPU; every stage's halo exchange overlaps the
        # interior tiles of its x pass (pcl_sharp_bc_stage)
        dims = parallel.proc_grid([args.nx, args.ny], size) if size > 1 else [1, 1]
        nxg, nyg = args.nx * dims[0], args.ny * dims[1]
        claw = build_sharp(nxg, nyg, args.math, args.dq_src)
    elif args.glob is not None:
        # strong scaling: the global grid is fixed and cut into px x py blocks (PETSc DMDA rule, parallel.py)
        scaling = "strong"
        nxg, nyg = args.glob
        dims = parallel.proc_grid([nxg, nyg], size) if size > 1 else [1, 1]
        claw = build(nxg, nyg, args.math, args.unsplit)
        args.nx, args.ny = claw.solution.state.q.shape[1:3]          # THIS rank's block (rank 0 prints)
    else:
        dims = parallel.proc_grid([args.nx, args.ny], size) if size > 1 else [1, 1]
        nxg, nyg = args.nx * dims[0], args.ny * dims[1]
        claw = build(nxg, nyg, args.math, args.unsplit)
    if args.state != "bubble":
        if args.ndim != 2 or args.solver != "classic" or args.app != "bubble" or size != 1:
            sys.stderr.write("bench.py --state applies to the 2-D Euler classic step on one GPU\n")
            sys.exit(2)
        claw = dense_state(claw) if args.state == "dense" else developed_state(claw)
        args.no_states = True
    elapsed, ms, nl, finite = timed_run(claw, args.steps, args.warmup)
    attempted = timed_run.attempted

    cells_total = float(nxg) * float(nyg) * (float(args.nx * dims[2]) if args.ndim == 3 else 1.0)
    value = cells_total * args.steps / elapsed / 1e6
    headline = args.ndim == 2 and not args.unsplit and args.solver == "classic" and args.app == "bubble"

    if rank == 0:
        ns = "pcl::%s::" % args.math
        names = [ns + "sweep_kernel<Euler5, 1> (x pass)", ns + "sweep_kernel<Euler5, 2> (y pass)"]
        if args.unsplit:
            names = [ns + "unsplit_x_kernel<Euler5> (x phase)", ns + "unsplit_ym_kernel<Euler5> (y phase, marching)"]
        avg = [ms[k] / max(1, nl[k]) for k in range(3)]
        dom = int(np.argmax(avg[:2]))
        bytes_launch = BYTES_PER_CELL_SWEEP * float(args.nx) * float(args.ny)
        bytes_pass = bytes_launch
        # a single block of the dim-split Euler step runs as ONE kernel (classic_fused.hpp): see pass_view
        one_kernel = headline and nl[2] > 0 and nl[2] >= nl[0]
        forms = dict(getattr(timed_run, "forms", {}))
        if one_kernel:
            names = [ns + "step2ds_kernel<Euler5> (x and y sweeps of the step in one kernel)",
                     ns + "sweep_kernel<Euler5, 1|2> (x pass, y pass: the other form of the step, trial steps)"]
            avg = [avg[2], max(avg[0], avg[1]) if nl[0] + nl[1] > 0 else None, 0.0]    # None: no trial step was sampled
            nl = [nl[2], nl[0] + nl[1], 0]
            dom = 0
        if args.unsplit and args.ndim == 2:
            # x phase: qold in, t1 out (80 B per cell); y phase: qold and t1 in, t1 out (120 B per cell)   
            bytes_launch = [80.0, 120.0][dom] * float(args.nx) * float(args.ny)
        if args.ndim == 3:
            # per directional sweep: read q (4) + aux (2), write q (4) doubles per cell
            names = [ns + "sweep3_kernel<VcAcoustics3D, 1> (x sweep)", ns + "sweep3_kernel<VcAcoustics3D, 2|3> (y, z sweeps)"]
            bytes_launch = (4 + 2 + 4) * 8 * float(args.nx) ** 3
            if args.unsplit:
                # per direction ONE marching kernel (classic3.hpp): qold + aux in, the accumulated state in and out
                # (algorithmic: q in, aux in, q out = (4 + 2 + 4) doubles per cell, as for the dim-split sweep)
                names = [ns + "march3p_kernel<VcAcoustics3D, 1> (x direction)",
                         ns + "march3p_kernel<VcAcoustics3D, 2|3> (y, z directions)"]
        if args.solver == "sharpclaw":
            # x pass: read the stage (5), write dq (5); y pass: read the stage, dq and the RK operand, write the result
            names = [ns + "sharp_kernel<Euler5, 1> (x pass of one RK stage)",
                     ns + "sharp_kernel<Euler5, 2> (y pass + fused RK combination)"]
            per = [80.0, 160.0]
            bytes_launch = per[dom] * float(args.nx) * float(args.ny)
        if args.app == "sphere":
            # every launch reads q (4) and the 16 aux planes and writes q or dq (4); the SharpClaw y pass also reads dq
            # and the RK operand
            if args.solver == "sharpclaw":
                names = [ns + "sharp_kernel<ShallowSphere, 1> (x pass of one RK stage)",
                         ns + "sharp_kernel<ShallowSphere, 2> (y pass + fused RK combination)"]
                per = [(4 + 16 + 4) * 8.0, (4 + 16 + 4 + 8) * 8.0]
            else:
                names = [ns + "unsplit_x_kernel<ShallowSphere, capa> (x phase)",
                         ns + "unsplit_y_kernel<ShallowSphere, capa> (y phase)"]
                per = [(4 + 16 + 4) * 8.0, (4 + 4 + 16 + 4) * 8.0]
            bytes_launch = per[dom] * float(args.nx) * float(args.ny)
        achieved = bytes_launch / (avg[dom] * 1e-3) / 1e9 if avg[dom] > 0 else 0.0
        # the committed PMC pass of the same command, for the default size of each line (tools/profile_r03.sh)
        if args.ndim == 3:
            pm = ("3d_unsplit", ["march3p_kernel", "VcAcoustics3D, %s," % ("1" if dom == 0 else "2")]) if args.unsplit else \
                 ("3d_dimsplit", ["sweep3_kernel", "VcAcoustics3D, %s" % ("1" if dom == 0 else "2")])
            dflt = size == 1 and args.math == "exact" and args.nx == (256 if args.unsplit else 512)
        elif args.app == "sphere":
            pm = ("sphere_sharpclaw", ["sharp_kernel", "ShallowSphere, %d" % (dom + 1)]) if args.solver == "sharpclaw" else \
                 ("sphere_classic", ["unsplit_x_kernel" if dom == 0 else "unsplit_y_kernel", "ShallowSphere"])
            dflt = size == 1 and args.math == "exact" and (nxg, nyg) == (2048, 1024)
        elif args.solver == "sharpclaw":
            pm = ("sharpclaw", ["sharp_kernel", "Euler5, %d" % (dom + 1)])
            dflt = size == 1 and args.math == "exact" and (args.nx, args.ny) == (4096, 4096) and not args.dq_src
        elif args.unsplit:
            pm = ("unsplit", ["unsplit_x_kernel" if dom == 0 else "unsplit_ym_kernel", "Euler5"])
            dflt = size == 1 and args.math == "exact" and (args.nx, args.ny) == (4096, 4096) and args.state == "bubble"
        else:
            key = {("exact", "bubble"): "exact", ("exact", "dense"): "exact_dense", ("fast", "dense"): "fast_dense"}.get(
                (args.math, args.state))
            if not one_kernel and key == "exact":
                key = "exact_twopass"           # PCL_TUNE_FUSED_STEP=0 / decomposed blocks without an interior box
            pm = (key, ["step2ds_kernel", "Euler5"] if one_kernel else ["sweep_kernel", "Euler5, %d," % (dom + 1)])
            dflt = size == 1 and key is not None and (args.nx, args.ny) == (4096, 4096)
        traffic, traffic_source = pmc_traffic(pm[0], pm[1], dflt)
        if scaling == "strong":
            grid_note = ("GLOBAL grid %dx%d fixed (strong scaling), %dx%d blocks, %dx%d cells on rank 0"
                         % (nxg, nyg, dims[0], dims[1], args.nx, args.ny))
        else:
            grid_note = "%dx%d cells per GPU (weak scaling)" % (args.nx, args.ny)
        out = {
            "metric": "Mcell*steps/s, 2-D Euler classic dim-split step (+ achieved HBM GB/s in roofline)",
            "value": value, "unit": "Mcell*steps/s", "n_gpus": size, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": ("test/acoustics/3d 3-D variable-coefficient acoustics, %d^3 cells, classic %s, MC "
                                    "limiter, order 2" % (args.nx, "UNSPLIT (step3, order_trans 22)" if args.unsplit
                                                          else "dim-split (step3ds)")) if args.ndim == 3 else
                                   "apps/euler 2D shock-bubble, %s, classic %s, "
                                   "mthlim=[4,4,4,4,2], order 2, source off"
                                   % (grid_note, "UNSPLIT order_trans=2" if args.unsplit else "dim-split"),
                       "global_grid": [nxg, nyg] + ([args.nx * dims[2]] if args.ndim == 3 else []), "proc_grid": dims, "math": ("exact (no FMA, IEEE div/sqrt; bit-identical to the reference)" if args.math == "exact"
                                else "strict (exact + IEEE quotients for underflow-range numerators)" if args.math == "strict"
                                else "fast (FMA contraction, reciprocal-multiply division; rtol 1e-12 vs reference)"),
                       "halo_transport": getattr(claw.solver, "halo_transport", "none (one block)"),
                       "exchange_ahead": bool(getattr(claw.solver, "exchange_ahead", False)),
                       "launches_timed": {names[0]: int(nl[0]), names[1]: int(nl[1])},
                       "steps_incl_rejected": int(attempted), "result_finite": finite},
            "roofline": {"bound": "hbm", "kernel": names[dom], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "avg_ms": {names[0]: avg[0], names[1]: avg[1]},
                         "algorithmic_bytes_per_launch": bytes_launch},
        }
        if one_kernel:
            # SURVEY 8(d)'s count (one read + one write of q per DIRECTIONAL PASS = 160 B per cell and step) describes
            # the two-pass algorithm; this launch does both passes on ONE read + ONE write of q.  roofline.achieved /
            # frac price the launch by what it has to move (80 B per cell); the 8(d) figure is kept beside it because
            # north_star's target is stated in it (60 % of 8 TB/s at 160 B per cell and step = 30 Gcell*steps/s)
            s8 = 2.0 * bytes_pass / (avg[0] * 1e-3) / 1e9 if avg[0] else 0.0
            out["roofline"]["algorithmic_bytes_note"] = (
                "one read + one write of q per STEP (80 B per cell): this launch does the x and the y sweeps of the step "
                "on a tile in LDS, so that is all it has to move; with its halo re-reads it moves 96.9 B per cell (see "
                "traffic)")
            out["roofline"]["survey_8d"] = {
                "bytes_per_cell_step": 2.0 * BYTES_PER_CELL_SWEEP, "achieved": s8, "unit": "GB/s",
                "frac": s8 / HBM_PEAK_GBS,
                "note": "SURVEY 8(d) counts one read + one write of q per directional pass (160 B per cell and step): "
                        "the traffic of the x pass + y pass form, which this kernel no longer moves -- the figure can "
                        "exceed 1 and is a throughput in the target's unit (north_star: 60 % = 30 Gcell*steps/s), not a "
                        "statement about the memory system; rounds 1-2 and the two-pass lines report it as roofline.frac"}
            out["config"]["step_form"] = dict(forms, note="two forms of the dim-split step with identical results; the "
                                              "solver runs the faster one, re-measured every 256 steps (3 timed steps of each)")
        ceil, ceil_src = copy_ceiling()
        if ceil:        # documentary: the fraction of what a plain copy reaches; `frac` stays against the 8 TB/s peak
            out["roofline"]["copy_ceiling"] = {"GB/s": ceil, "frac_of_ceiling": achieved / ceil, "source": ceil_src}
        if args.solver == "sharpclaw":
            out["metric"] = "Mcell*steps/s, 2-D Euler SharpClaw WENO5 + SSP104 step (10 right-hand sides per step)"
            out["config"]["workload"] = ("apps/euler 2D shock-bubble, %dx%d cells, SharpClaw lim_type=2 (WENO5), "
                                         "SSP104, %s" % (args.nx, args.ny,
                                                         "dq_src = dq_Euler_radial added by the last pass of every stage"
                                                         if args.dq_src else "source off"))
        if args.app == "sphere":
            out["metric"] = ("Mcell*steps/s, shallow water on the sphere, %s"
                             % ("SharpClaw WENO5 + SSP104 step (10 right-hand sides per step)"
                                if args.solver == "sharpclaw" else
                                "classic unsplit step2qcor step + Strang-split Coriolis source"))
            out["config"]["workload"] = (
                "apps/shallow-sphere Rossby-Haurwitz wave, %dx%d cells, 16 aux planes, capacity function, %s"
                % (args.nx, args.ny, "SharpClaw lim_type=2 (WENO5), SSP104 (BASELINE configs[4] on one GPU)"
                   if args.solver == "sharpclaw" else
                   "classic rpn2/rpt2_shallow_sphere, order_trans=2, MC limiter, src_split=2 (the reference app)"))
        if args.state != "bubble":
            out["config"]["workload"] += "; STATE: %s (not the shock-bubble initial condition)" % args.state
        elif headline:
            out["config"]["state_note"] = (
                "headline = shock-bubble initial condition (BASELINE configs[2]): mostly undisturbed gas; wavefronts "
                "without a jump take an exact shortcut and absent wave families skip the limiter (bit-identical "
                "results).  The kernels' cost depends on the state: see dense_state (input-independent figure) and "
                "developed_state in this same line")
        if headline and size == 1 and args.math == "exact" and not args.no_states:
            desc_dense = "rho=1+.1U, mx=.1U, my=.05U, E=2.5+.1U, tracer=U (U uniform random per cell): a jump at every interface, all five wave families everywhere"
            desc_dev = ("the reference regression's t=0.2 solution (test/euler/2d/shockbubble.py, 160x40, shock through "
                        "the bubble) interpolated bilinearly to this grid")
            k2 = args.steps if args.steps <= 300 else 300      # these objects are rates; 300 steps is > 0.25 s
            out["dense_state"] = state_object("dense", dense_state(build(nxg, nyg, "exact", False)), k2, args.warmup,
                                              cells_total, bytes_pass, desc_dense)
            out["developed_state"] = state_object("developed", developed_state(build(nxg, nyg, "exact", False)), k2,
                                                  args.warmup, cells_total, bytes_pass, desc_dev)
            # "source term off for the pure classic-step figure, on for the app figure": the same state
            # with the app's axisymmetric source term (step_Euler_radial, device version) after every step
            el5, ms5, nl5, fin5 = timed_run(build(nxg, nyg, "exact", False, with_src=True), k2, args.warmup)
            out["app_figure"] = {"value": cells_total * k2 / el5 / 1e6, "unit": "Mcell*steps/s", "steps": k2,
                                 "ms_per_step": el5 / k2 * 1e3, "math": "exact",
                                 "what": "classic dim-split step + the app's radial source term (Godunov splitting, "
                                         "EulerRadialSource on the device: one more read + write of q per step)",
                                 "result_finite": fin5}
            if not args.no_app_run:
                out["app_run"] = app_run_object(nxg, nyg, cells_total)
            if args.steps < 200:
                # a K-step region of ~10 ms is thin: the same headline state for 1000 steps
                el4, ms4, nl4, fin4 = timed_run(build(nxg, nyg, "exact", False), 1000, args.warmup)
                lab4, dur4, byt4 = pass_view(ms4, nl4, bytes_pass)
                out["sustained"] = {"steps": 1000, "value": cells_total * 1000 / el4 / 1e6, "unit": "Mcell*steps/s",
                                    "ms_per_step": el4 / 1000 * 1e3, "timed_region_s": el4,
                                    "roofline_frac": byt4 / (dur4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                    "roofline_frac_whole_step": 2.0 * bytes_pass / (el4 / 1000) / 1e9 / HBM_PEAK_GBS,
                                    "avg_ms": lab4, "result_finite": fin4}
        if size == 1 and headline and args.math == "exact" and (args.extras or not args.no_states):
            # the second arithmetic mode (FMA contraction, reciprocal-multiply division, one-step Newton sqrt;
            # tests/test_gpu_apps.py holds it to the north-star tolerance rtol 1e-12 on the reference goldens) on the
            # headline state and on the dense state
            k3 = args.steps if args.steps <= 300 else 300
            fm = {"parity": "rtol 1e-12 vs the reference goldens (not bit-identical)"}
            for tag, mk in (("bubble", lambda c: c), ("dense", dense_state)):
                el2, ms2, nl2, fin2 = timed_run(mk(build(nxg, nyg, "fast", False)), k3, args.warmup)
                lab2, dur2, byt2 = pass_view(ms2, nl2, bytes_pass)
                fm[tag] = {"value": cells_total * k3 / el2 / 1e6, "unit": "Mcell*steps/s", "steps": k3,
                           "ms_per_step": el2 / k3 * 1e3,
                           "roofline_frac": byt2 / (dur2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "roofline_frac_whole_step": 2.0 * bytes_pass / (el2 / k3) / 1e9 / HBM_PEAK_GBS,
                           "avg_ms": lab2, "result_finite": fin2}
            out["fast_math"] = fm
        if args.ndim == 3:
            out["metric"] = ("Mcell*steps/s, 3-D acoustics classic %s step (+ achieved HBM GB/s in roofline)"
                             % ("UNSPLIT (step3, order_trans 22)" if args.unsplit else "dim-split (step3ds)"))
        elif args.unsplit and args.solver == "classic" and args.app == "bubble":
            out["metric"] = "Mcell*steps/s, 2-D Euler classic UNSPLIT step (step2, order_trans 2) (+ achieved HBM GB/s in roofline)"
        if size == 1 and not args.no_cpu_baseline and args.ndim == 2 and args.solver == "classic" and args.app == "bubble":
            try:
                out["cpu_baseline"] = cpu_baseline(args.nx, args.ny)
            except Exception as e:      # the oracle is optional infrastructure, never the product
                out["cpu_baseline"] = {"value": None, "unit": "Mcell*steps/s", "cores": 1, "kind": "port",
                                       "sample": "failed: %s" % e}
            try:
                out["cpu_baseline"]["all_cores"] = cpu_baseline_all_cores(args.nx, args.ny)
            except Exception as e:
                out["cpu_baseline"]["all_cores"] = {"value": None, "sample": "failed: %s" % e}
        print(json.dumps(out))
    parallel.shutdown()


if __name__ == "__main__":
    main()
