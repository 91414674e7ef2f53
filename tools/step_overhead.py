#!/usr/bin/env python
"""tools/step_overhead.py -- wall time per accepted classic step vs grid size: separates the fixed
per-step cost (BC launches, CFL read-back, host logic) from the sweep kernels."""
import os, sys, time, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyclaw_amd as pyclaw
from pyclaw_amd import _lib
from apps import problems
for n in (64, 256, 1024, 4096):
    claw = problems.shockbubble(pyclaw, mx=n, my=n, device_callbacks=True, with_src=False,
                                dt_initial=0.005 * (2.0 / n) / (2.0 / 160.0), run=False)
    s, sol = claw.solver, claw.solution
    s.setup(sol); s.dt = s.dt_initial
    s.begin_resident(sol)
    for _ in range(10): s.evolve_to_time(sol)
    L = _lib.lib(); _lib.check(L.pcl_kernel_timing(s._h, 1)); _lib.check(L.pcl_sync(s._h))
    K = 50
    t0 = time.perf_counter()
    for _ in range(K): s.evolve_to_time(sol)
    _lib.check(L.pcl_sync(s._h)); el = time.perf_counter() - t0
    ms = np.zeros(2); nl = np.zeros(2, dtype=np.int64)
    _lib.check(L.pcl_kernel_timing_read(s._h, _lib.d(ms), nl.ctypes.data_as(ctypes.POINTER(ctypes.c_long))))
    s.end_resident(sol); s.teardown()
    print(json.dumps({"n": n, "us_per_step": el / K * 1e6, "kernels_us": (ms[0] + ms[1]) / nl[0] * 1e3,
                      "overhead_us": el / K * 1e6 - (ms[0] + ms[1]) / nl[0] * 1e3}))
