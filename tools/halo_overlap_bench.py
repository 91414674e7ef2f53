#!/usr/bin/env python
"""Time pcl_bc_step at 4096^2 Euler on ONE GPU whose 8 halo neighbours are itself (RCCL send/recv to self):
no comm / sequential exchange (PCL_HALO_OVERLAP=0) / exchange overlapped with the interior x tiles (=1) / the same with
the exchange sent ahead, behind the previous step's y pass (pcl_halo_exchange_ahead).  PCL_HALO_BENCH_STATE=bubble|dense:
the shock-bubble initial condition / the bench's dense state instead of the nearly uniform default.
Run each mode in its own process: the env knob is read at pcl_comm_init.
  python tools/halo_overlap_bench.py [nx ny steps]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyclaw_amd import _lib as L


def run(mx, my, steps, comm, overlap, ahead=False, state="uniform"):
    os.environ["PCL_HALO_OVERLAP"] = str(overlap)
    lib = L.lib()
    cfg = L.Config()
    cfg.ndim = 2
    cfg.n[0], cfg.n[1] = mx, my
    cfg.mbc = 2
    cfg.meqn, cfg.mwaves, cfg.rp = 5, 5, 11
    cfg.method[1], cfg.method[2] = 2, -1
    if os.environ.get("PCL_HALO_BENCH_UNSPLIT"):         # the unsplit step (order_trans = 2) instead of the dim-split one
        cfg.method[2] = 2
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
        if ahead:
            yes = C.c_int(0)
            L.check(lib.pcl_halo_can_overlap(h, C.byref(yes)))
            L.check(lib.pcl_halo_exchange_ahead(h, yes.value))       # 2: the one-kernel step's order (rim tiles first)
    else:
        bc = np.full(4, 2, dtype=np.int32)
    rng = np.random.default_rng(0)
    q = np.empty((5, mx, my), order="F")
    if state == "bubble":      # the shock-bubble initial condition of BASELINE configs[2..3] (apps/problems.py sb_qinit)
        x = (np.arange(mx) + 0.5) * (2.0 / mx)
        y = (np.arange(my) + 0.5) * (0.5 / my)
        r = np.sqrt((x[:, None] - 0.5) ** 2 + (y[None, :] - 0.0) ** 2)
        q[0] = 0.1 * (r <= 0.2) + 1.0 * (r > 0.2); q[1] = 0.0; q[2] = 0.0; q[3] = 1.0 / 0.4; q[4] = 1.0 * (r <= 0.2)
    elif state == "dense":     # the bench's dense state: a jump at every interface, every wave family everywhere
        q[0] = 1.0 + 0.1 * rng.random((mx, my)); q[1] = 0.1 * rng.random((mx, my)); q[2] = 0.05 * rng.random((mx, my))
        q[3] = 2.5 + 0.1 * rng.random((mx, my)); q[4] = rng.random((mx, my))
    else:
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
    only = sys.argv[4] if len(sys.argv) > 4 else None       # "none" | "seq" | "ovl" | "split": one mode (profiler runs)
    state = os.environ.get("PCL_HALO_BENCH_STATE", "uniform")      # uniform | bubble | dense
    for tag, label, comm, ov in (("none", "no comm, local periodic BC", False, 0), ("seq", "self-halo sequential", True, 0),
                                 ("ovl", "self-halo overlapped", True, 1),
                                 ("ahead", "self-halo overlapped + exchange-ahead", True, 1),
                                 ("split", "split launches, one stream (mode 2)", True, 2)):
        if only and only != tag:
            continue
        ms, cfl = run(mx, my, steps, comm, ov, ahead=(tag == "ahead"), state=state)
        print("%-38s %.4f ms/step  cfl %.6f" % (label, ms, cfl), flush=True)
