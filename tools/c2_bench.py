#!/usr/bin/env python
"""tools/c2_bench.py -- BASELINE configs[1]: apps/acoustics 2D, 1024 x 1024, classic unsplit (rpn2/rpt2) + MC limiter.
Prints Mcell*steps/s and the per-phase kernel time (x phase / y phase of the unsplit step)."""
import ctypes, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyclaw_amd as pyclaw
from pyclaw_amd import _lib
from apps import problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
for split in (0, 1):
    claw = problems.acoustics2D(pyclaw, mx=n, my=n, dim_split=split, run=False)
    s, sol = claw.solver, claw.solution
    s.setup(sol); s.dt = s.dt_initial
    L = _lib.lib()
    s.begin_resident(sol)
    for _ in range(10):
        s.evolve_to_time(sol)
    _lib.check(L.pcl_kernel_timing(s._h, 1)); _lib.check(L.pcl_sync(s._h))
    t0 = time.perf_counter()
    for _ in range(steps):
        s.evolve_to_time(sol)
    _lib.check(L.pcl_sync(s._h)); el = time.perf_counter() - t0
    ms = np.zeros(2); nl = np.zeros(2, dtype=np.int64)
    _lib.check(L.pcl_kernel_timing_read(s._h, _lib.d(ms), nl.ctypes.data_as(ctypes.POINTER(ctypes.c_long))))
    s.end_resident(sol); s.teardown()
    print(json.dumps({"config": "acoustics 2D %dx%d classic %s MC" % (n, n, "dim-split" if split else "unsplit rpn2/rpt2"),
                      "Mcell_steps_per_s": n * n * steps / el / 1e6, "us_per_step": el / steps * 1e6,
                      "x_us": ms[0] / max(1, nl[0]) * 1e3, "y_us": ms[1] / max(1, nl[1]) * 1e3}), flush=True)
