#!/usr/bin/env python
"""tools/sharp_bench.py -- throughput of the SharpClaw path (WENO5 + SSP104) on one GPU:
2-D Euler shock-bubble state, nx x ny cells, K steps.  Prints Mcell*steps/s and per-stage kernel time."""
import argparse, ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyclaw_amd as pyclaw
from pyclaw_amd import _lib
from apps import problems

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=2048)
ap.add_argument("--ny", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--math", default="exact")
ap.add_argument("--lim", type=int, default=2)
args = ap.parse_args()

x = pyclaw.Dimension('x', 0.0, 2.0, args.nx)
y = pyclaw.Dimension('y', 0.0, 0.5, args.ny)
state = pyclaw.State(pyclaw.Grid([x, y]), 5, 1)
state.aux_global['gamma'] = problems.gamma
state.aux_global['gamma1'] = problems.gamma1
problems.sb_qinit(state)
problems.sb_auxinit(state)
solver = pyclaw.SharpClawSolver2D()
solver.rp = pyclaw.riemann.rp_euler_5wave_2d
solver.mwaves = 5
solver.lim_type = args.lim
solver.math = args.math
solver.cfl_max, solver.cfl_desired = 2.5, 2.45
rinf, vinf, einf = problems.shock_state()
solver.user_bc_lower = pyclaw.ConstantStateBC([rinf, rinf * vinf, 0., einf, 0.])
solver.bc_lower = [pyclaw.BC.custom, pyclaw.BC.reflecting]
solver.bc_upper = [pyclaw.BC.outflow, pyclaw.BC.outflow]
solver.aux_bc_lower = [pyclaw.BC.outflow] * 2
solver.aux_bc_upper = [pyclaw.BC.outflow] * 2
solver.dt_initial = 0.4 * (2.0 / args.nx)
sol = pyclaw.Solution(state)
solver.setup(sol)
solver.dt = solver.dt_initial
L = _lib.lib()
solver.begin_resident(sol)
for _ in range(3):
    solver.evolve_to_time(sol)
_lib.check(L.pcl_kernel_timing(solver._h, 1))
_lib.check(L.pcl_sync(solver._h))
t0 = time.perf_counter()
for _ in range(args.steps):
    solver.evolve_to_time(sol)
_lib.check(L.pcl_sync(solver._h))
el = time.perf_counter() - t0
ms = np.zeros(2); nl = np.zeros(2, dtype=np.int64)
_lib.check(L.pcl_kernel_timing_read(solver._h, _lib.d(ms), nl.ctypes.data_as(ctypes.POINTER(ctypes.c_long))))
solver.end_resident(sol)
print(json.dumps({"solver": "SharpClaw WENO5+SSP104 (10 stages/step)", "grid": [args.nx, args.ny], "math": args.math,
                  "lim_type": args.lim, "Mcell_steps_per_s": args.nx * args.ny * args.steps / el / 1e6,
                  "ms_per_step": el / args.steps * 1e3, "x_kernel_ms": ms[0] / nl[0], "y_kernel_ms": ms[1] / nl[1],
                  "stage_evals": int(nl[0]), "cfl": solver.cfl.get_cached_max(),
                  "finite": bool(np.isfinite(sol.state.q).all())}))
