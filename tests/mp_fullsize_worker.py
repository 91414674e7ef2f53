"""
BASELINE configs[3] (C4: Euler 8192 x 8192 on 2 x 4 blocks) and configs[4] (C5: SharpClaw WENO5 on the sphere,
2048 x 1024 on 4 blocks) AT THEIR SIZES through the decomposed device path: builders shared by
tests/test_gpu_c4c5.py (serial run on one GPU + oracle windows) and by the worker processes it launches
(`python tests/mp_fullsize_worker.py <case>`; 4 ranks sharing the box's one GPU over the host-staged wire, like
tests/mp_gpu_worker.py).  A worker hashes its block of the final state and compares it with the hash the serial run
left for the same index range (PCL_FS_EXPECT = path of a small JSON file): decomposed == serial, bit for bit, which is
the reference's own acceptance test for its parallel layer (test/test_examples.py:264-277, at 1e-14 there).

Cases
  c4_periodic   8192 x 8192, 2 x 2 blocks of 4096^2; Euler 5-wave, dim-split, mthlim 4,4,4,4,2, the app's radial
                source fused into the y pass; a synthetic state with a jump at every interface plus constant patches,
                periodic sides (every block has all eight neighbours, corners included); 3 fixed-dt steps
  c4_layout     8192 x 4096, 2 x 2 blocks of 4096 x 2048 = the block shape of C4's 2 x 4 layout; otherwise the same
  c4_unsplit    the c4_layout grid with the unsplit algorithm (order_trans 2): the transverse terms read corner ghosts
  c4_app        the shock-bubble app itself on 8192 x 8192 (inflow / reflecting / outflow sides, adaptive dt, source)
  c5_sphere     shallow water on the sphere, SharpClaw WENO5 + SSP104, 2048 x 1024 as 1 x 4 blocks of 2048 x 256
                (mirrored pole boundary, mbc = 3, 16 aux planes, capacity function); 1 fixed-dt step = 10 stages

The synthetic state uses only integer arithmetic and + - * / on doubles, so every process computes the same bits for
a cell whatever block it sits in.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GAMMA = 1.4
GAMMA1 = GAMMA - 1.0          # 0.39999999999999991: the reference scripts compute it (test/euler/2d/shockbubble.py:7-8)

CASES = {
    #  name         (nx,   ny,   proc grid, steps)
    "c4_periodic": (8192, 8192, "2x2", 3),
    "c4_layout": (8192, 4096, "2x2", 3),
    "c4_unsplit": (8192, 4096, "2x2", 2),
    "c4_app": (8192, 8192, "2x2", None),
    "c5_sphere": (2048, 1024, "1x4", 1),
}
# small twins of the same builders for the CPU-side self-checks and quick GPU runs (PCL_FS_SMALL=1)
SMALL = {"c4_periodic": (256, 192), "c4_layout": (256, 128), "c4_unsplit": (256, 128), "c4_app": (320, 80),
         "c5_sphere": (128, 64)}


def case_shape(case):
    nx, ny, pg, steps = CASES[case]
    if os.environ.get("PCL_FS_SMALL") == "1":
        nx, ny = SMALL[case]
    return nx, ny, pg, steps


def synth_euler(i, j):
    """Euler state (5, len(i), len(j)) at GLOBAL cell indices i, j: white noise of 1/1024 granularity around a
    quiescent gas (the bench's dense state: a jump at every interface), two constant patches (jump-free wavefronts) and
    a strong step; integer hashing + exact double arithmetic only."""
    I = np.asarray(i, dtype=np.int64)[:, None]
    J = np.asarray(j, dtype=np.int64)[None, :]

    def u(m):
        h = (I * 73856093) ^ (J * 19349663) ^ (m * 83492791)
        h = (h ^ (h >> 13)) * 1274126177
        return ((h ^ (h >> 16)) & 1023).astype(np.float64) / 1024.0
    q = np.empty((5, I.shape[0], J.shape[1]), order="F")
    q[0] = 1.0 + 0.1 * u(0)
    q[1] = 0.1 * u(1) - 0.03
    q[2] = 0.05 * u(2) - 0.02
    q[3] = 2.5 + 0.1 * u(3)
    q[4] = u(4)
    patch = ((I // 160) % 5 == 2) & ((J // 96) % 7 == 3)          # constant patches, some across block faces
    for m, v in enumerate((0.3, 0.2, -0.1, 1.1, 1.0)):
        q[m] = np.where(patch, v, q[m])
    step = ((I // 1024) + (J // 1024)) % 3 == 1                    # strong steps along lines that include the faces
    for m, f in enumerate((2.0, 2.0, 2.0, 3.0, 1.0)):
        q[m] = np.where(step & ~patch, q[m] * f, q[m])
    return q


def fixed_dt(case):
    nx, ny, _, _ = case_shape(case)
    return 0.05 / max(nx, ny)        # Courant number about 0.3 on the synthetic state (speeds up to ~3)


def build(case, pyclaw):
    """the Controller of a case, not yet run (block-local data when the process group has several ranks)"""
    from apps import problems
    nx, ny, _, steps = case_shape(case)
    if case == "c4_app":
        tfinal = 4.0e-4 * 160.0 / nx * 10            # a handful of adaptive steps from dt_initial ~ dx
        return problems.shockbubble(pyclaw, mx=nx, my=ny, tfinal=tfinal, device_callbacks=True,
                                    dt_initial=0.005 * 160.0 / nx, run=False)
    if case == "c5_sphere":
        from apps import shallow_sphere as S
        claw = S.shallow_sphere(pyclaw, nx, ny, run=False, solver_type='sharpclaw', nout=1)
        dt = 0.4 * min(4.0 / nx, 2.0 / ny) / 4.0     # Courant number ~1 for sqrt(g h) ~ 3.3 plus the flow
        claw.solver.dt_variable = False
        claw.solver.dt_initial = dt
        claw.tfinal = steps * dt
        return claw
    x = pyclaw.Dimension('x', 0.0, 2.0, nx)
    y = pyclaw.Dimension('y', 0.0, 2.0 * ny / nx, ny)
    grid = pyclaw.Grid([x, y])
    state = pyclaw.State(grid, 5, 1)
    state.aux_global['gamma'] = GAMMA
    state.aux_global['gamma1'] = GAMMA1
    (i0, i1), (j0, j1) = [(d.nstart, d.nend) for d in grid.dimensions]
    state.q[...] = synth_euler(np.arange(i0, i1), np.arange(j0, j1))
    problems.sb_auxinit(state)                       # aux[0] = y of the cell centre: the radial coordinate
    solver = pyclaw.ClawSolver2D()
    solver.rp = pyclaw.riemann.rp_euler_5wave_2d
    solver.mwaves = 5
    solver.limiters = [4, 4, 4, 4, 2]
    solver.dim_split = case != "c4_unsplit"
    solver.order_trans = 2
    solver.src_split = 1
    solver.step_src = pyclaw.EulerRadialSource(GAMMA1, 2)
    solver.cfl_max, solver.cfl_desired = 1.0, 0.9
    solver.dt_variable = False
    solver.dt_initial = fixed_dt(case)
    for k in range(2):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.periodic
        solver.aux_bc_lower[k] = solver.aux_bc_upper[k] = pyclaw.BC.outflow
    claw = pyclaw.Controller()
    claw.keep_copy = False
    claw.output_format = None
    claw.tfinal = steps * solver.dt_initial
    claw.nout = 1
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    return claw


def block_hash(a):
    """hex digest of an array's values in Fortran order"""
    raw = np.asfortranarray(a).tobytes(order="F") if a.nbytes < (1 << 20) else None
    try:
        import xxhash
        h = xxhash.xxh3_128()
    except ImportError:
        import hashlib
        h = hashlib.blake2b(digest_size=16)
    if raw is not None:
        h.update(raw)
    else:
        af = np.asfortranarray(a)
        flat = af.reshape(-1, order="F")
        step = 1 << 24
        for s in range(0, flat.size, step):
            h.update(flat[s:s + step].tobytes())
    return h.hexdigest()


def run_case(case, pyclaw):
    """run the case's Controller; returns (final q of this process' block, its index ranges, status)"""
    claw = build(case, pyclaw)
    claw.keep_copy = False
    claw.output_format = None
    status = claw.run()
    st = claw.solution.state
    rng = [(int(d.nstart), int(d.nend)) for d in st.grid.dimensions]
    out = {"numsteps": int(claw.solver.status["numsteps"]), "cflmax": repr(float(claw.solver.status["cflmax"])),
           "dt": repr(float(claw.solver.dt)), "t": repr(float(claw.solution.t)),
           "exchange_ahead": bool(getattr(claw.solver, "exchange_ahead", False))}
    q = st.q
    claw.solver.teardown()
    return q, rng, out


def main():
    case = sys.argv[1]
    nx, ny, pg, _ = case_shape(case)
    os.environ["PCL_PROC_GRID"] = pg
    import pyclaw_amd as pyclaw
    from pyclaw_amd import parallel
    parallel.init()
    rank, size = parallel.rank(), parallel.world_size()
    q, rng, out = run_case(case, pyclaw)
    with open(os.environ["PCL_FS_EXPECT"]) as f:
        expect = json.load(f)
    key = "%d:%d,%d:%d" % (rng[0][0], rng[0][1], rng[1][0], rng[1][1])
    ok = True
    msgs = []
    if key not in expect["blocks"]:
        ok = False
        msgs.append("block %s is not one of the expected %s" % (key, sorted(expect["blocks"])))
    else:
        got = block_hash(q)
        if got != expect["blocks"][key]:
            ok = False
            msgs.append("block %s: hash %s != serial %s" % (key, got, expect["blocks"][key]))
    for k in ("numsteps", "cflmax", "dt", "t"):
        if out[k] != expect["status"][k]:
            ok = False
            msgs.append("%s: %s != serial %s" % (k, out[k], expect["status"][k]))
    if not np.isfinite(q).all():
        ok = False
        msgs.append("non-finite values in the block")
    oks = parallel.allgather({"rank": rank, "ok": ok, "key": key, "msgs": msgs})
    if rank == 0:
        keys = sorted(o["key"] for o in oks)
        distinct = len(set(keys)) == size and set(keys) == set(expect["blocks"])
        for o in oks:
            for m in o["msgs"]:
                print("rank %d: %s" % (o["rank"], m))
        print("case %s: %d ranks, blocks %s, steps %s, cflmax %s, exchange-ahead %s, bit-identical to the serial run: %s"
              % (case, size, keys, out["numsteps"], out["cflmax"], out["exchange_ahead"],
                 all(o["ok"] for o in oks) and distinct))
        ok = all(o["ok"] for o in oks) and distinct
    parallel.barrier()
    parallel.shutdown()
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    main()
