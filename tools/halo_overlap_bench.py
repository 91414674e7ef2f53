#!/usr/bin/env python
"""Time pcl_bc_step at 4096^2 Euler on ONE GPU whose 8 halo neighbours are itself (RCCL send/recv to self):
no comm / sequential exchange (PCL_HALO_OVERLAP=0) / exchange overlapped with the interior x tiles (=1).
Run each mode in its own process: the env knob is read at pcl_comm_init.
  python tools/halo_overlap_bench.py [nx ny steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyclaw_amd import _lib as L


def run(mx, my, steps, comm, overlap):
    os.environ["PCL_HALO_OVERLAP"] = str(overlap)
    lib = L.lib()
    cfg = L.Config()
    cfg.ndim = 2
    cfg.n[0], cfg.n[1] = mx, my
    cfg.mbc = 2
    cfg.meqn, cfg.mwaves, cfg.rp = 5, 5, 11
    cfg.method[1], cfg.method[2] = 2, -1
    for k, v in enumerate([4, 4, 4, 4, 2]):
        cfg.mthlim[k] = v
    cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.d[0], cfg.d[1] = 2.0 / mx, 0.5 / my
    h = C.c_void_p()
    L.check(lib.pcl_create(C.byref(cfg), C.byref(h)))
    if comm:
        uid = C.create_string_buffer(128)
        L.check(lib.pcl_comm_unique_id(uid))
        L.check(lib.pcl_comm_init(h, 1, 0, uid, L.i(np.zeros(8, dtype=np.int32))))
        bc = np.full(4, -1, dtype=np.int32)
    else:
        bc = np.full(4, 2, dtype=np.int32)
    rng = np.random.default_rng(0)
    q = np.empty((5, mx, my), order="F")
    q[0] = 1 + 0.1 * rng.random((mx, my)); q[1] = 0.1; q[2] = 0.0; q[3] = 2.5; q[4] = 0.0
    L.check(lib.pcl_put_q(h, L.d(q), 0))
    consts = np.zeros(32)
    cfl = C.c_double()
    dt = 0.1 * cfg.d[0]
    for _ in range(5):
        L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), dt, C.cast(C.byref(cfl), L.dp)))
    L.check(lib.pcl_sync(h))
    t0 = time.perf_counter()
    for _ in range(steps):
        L.check(lib.pcl_bc_step(h, L.i(bc), L.d(consts), dt, C.cast(C.byref(cfl), L.dp)))
    L.check(lib.pcl_sync(h))
    ms = (time.perf_counter() - t0) / steps * 1e3
    lib.pcl_destroy(h)
    return ms, cfl.value


if __name__ == "__main__":
    mx = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    my = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    for label, comm, ov in (("no comm, local periodic BC", False, 0), ("self-halo sequential", True, 0),
                            ("self-halo overlapped", True, 1), ("split launches, one stream (mode 2)", True, 2)):
        ms, cfl = run(mx, my, steps, comm, ov)
        print("%-38s %.4f ms/step  cfl %.6f" % (label, ms, cfl), flush=True)
