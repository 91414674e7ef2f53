#!/usr/bin/env python
"""x-pass time with the boundary conditions evaluated inside the kernel (pcl_bc_step, 'virtual ghost cells')
vs. separate ghost-fill launches + plain x pass (pcl_bc + pcl_step_hyperbolic), same shock-bubble state."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyclaw_amd as pyclaw
from pyclaw_amd import _lib as L
from apps import problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
claw = problems.shockbubble(pyclaw, mx=n, my=n, device_callbacks=True, with_src=False,
                            dt_initial=0.005 * (2.0 / n) / (2.0 / 160.0), run=False)
s, sol = claw.solver, claw.solution
s.setup(sol); s.dt = s.dt_initial
lib = L.lib(); h = s._h
s.begin_resident(sol)
spec = s._device_bc_spec(sol.state)
cfl = C.c_double(); cp = C.cast(C.byref(cfl), L.dp)
dt = s.dt_initial * 0.5
for mode in ("fused", "separate", "fused", "separate"):
    for rep in range(2):
        if rep == 1:
            L.check(lib.pcl_kernel_timing(h, 1))
        for _ in range(20):
            if mode == "fused":
                L.check(lib.pcl_bc_step(h, spec[2], spec[3], dt, cp))
            else:
                rinf, vinf, einf = problems.shock_state()
                st = np.array([rinf, rinf * vinf, 0., einf, 0.])
                L.check(lib.pcl_bc_const(h, 0, 0, L.d(st)))
                L.check(lib.pcl_bc(h, 0, 1, 1)); L.check(lib.pcl_bc(h, 1, 0, 3)); L.check(lib.pcl_bc(h, 1, 1, 1))
                L.check(lib.pcl_step_hyperbolic(h, dt, cp))
    ms = np.zeros(2); nl = np.zeros(2, dtype=np.int64)
    L.check(lib.pcl_kernel_timing_read(h, L.d(ms), nl.ctypes.data_as(C.POINTER(C.c_long))))
    L.check(lib.pcl_kernel_timing(h, 0))
    print("%-9s x %.4f ms  y %.4f ms  cfl %.4f" % (mode, ms[0] / nl[0], ms[1] / nl[1], cfl.value), flush=True)
