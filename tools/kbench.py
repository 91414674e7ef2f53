#!/usr/bin/env python
"""
tools/kbench.py -- kernel micro-benchmark: HIP-event time of the x / y sweep kernels on a
resident 2-D Euler state, several launches per variant, for A/B work on one box.

  python tools/kbench.py [--n 4096] [--reps 30] [--state bubble|random] [--math exact fast]

Tuning knobs are read by the library from the environment at first launch, so each variant
runs in a child process; variants are interleaved (round-robin) to average out DVFS drift.
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(args):
    import ctypes
    import numpy as np
    import pyclaw_amd as pyclaw
    from pyclaw_amd import _lib
    from apps import problems
    n = args.n
    claw = problems.shockbubble(pyclaw, mx=n, my=n, device_callbacks=True, with_src=False,
                                dt_initial=0.005 * (2.0 / n) / (2.0 / 160.0), run=False, math=args.math[0],
                                dim_split=not args.unsplit)
    solver, solution = claw.solver, claw.solution
    if args.capa:      # the aux plane (y coordinate) doubles as a smooth capacity function near 1
        solution.state.aux[0] = 1.0 + 0.2 * solution.state.aux[0]
        solution.state.mcapa = 0
    if args.state == "random":
        rng = np.random.default_rng(0)
        q = solution.state.q
        shape = q.shape[1:]
        q[0] = 1.0 + 0.1 * rng.random(shape)
        q[1] = 0.1 * rng.random(shape)
        q[2] = 0.05 * rng.random(shape)
        q[3] = 2.5 + 0.1 * rng.random(shape)
        q[4] = rng.random(shape)
    solver.setup(solution)
    solver.dt = solver.dt_initial
    L = _lib.lib()
    h = solver._h
    solver.begin_resident(solution)
    for _ in range(args.warm):
        solver.evolve_to_time(solution)
    _lib.check(L.pcl_kernel_timing(h, 1))
    for _ in range(args.reps):
        solver.evolve_to_time(solution)
    ms = np.zeros(2)
    nl = np.zeros(2, dtype=np.int64)
    _lib.check(L.pcl_kernel_timing_read(h, _lib.d(ms), nl.ctypes.data_as(ctypes.POINTER(ctypes.c_long))))
    print(json.dumps({"x_ms": ms[0] / nl[0], "y_ms": ms[1] / nl[1], "launches": int(nl[0])}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--warm", type=int, default=5)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--state", default="bubble")
    ap.add_argument("--math", nargs="+", default=["exact"])
    ap.add_argument("--env", nargs="*", default=[""], help="variants as K=V[,K=V] strings")
    ap.add_argument("--unsplit", action="store_true", help="step2.f (transverse solves) instead of step2ds.f")
    ap.add_argument("--capa", action="store_true", help="with a capacity function")
    ap.add_argument("--child", action="store_true")
    args = ap.parse_args()
    if args.child:
        child(args)
        return
    res = {}
    for r in range(args.rounds):
        for math in args.math:
            for var in args.env:
                env = dict(os.environ)
                for kv in filter(None, var.split(",")):
                    k, v = kv.split("=")
                    env[k] = v
                out = subprocess.check_output([sys.executable, os.path.abspath(__file__), "--child", "--n", str(args.n),
                                               "--reps", str(args.reps), "--warm", str(args.warm), "--state", args.state,
                                               "--math", math] + (["--unsplit"] if args.unsplit else []) +
                                              (["--capa"] if args.capa else []), env=env)
                d = json.loads(out.decode().strip().splitlines()[-1])
                res.setdefault((math, var), []).append(d)
    for (math, var), ds in res.items():
        xs = [d["x_ms"] for d in ds]
        ys = [d["y_ms"] for d in ds]
        print("%-6s %-28s x %.4f ms (min %.4f)   y %.4f ms (min %.4f)   GB/s x %.0f y %.0f" % (
            math, var or "-", sum(xs) / len(xs), min(xs), sum(ys) / len(ys), min(ys),
            80.0 * args.n * args.n / (sum(xs) / len(xs)) / 1e6, 80.0 * args.n * args.n / (sum(ys) / len(ys)) / 1e6))


if __name__ == "__main__":
    main()
