"""One-off probe: do states whose momenta / tracer sit in the underflow range (1e-300 .. 5e-324) keep the exact build
bit-identical to the oracle?  (The shared-reciprocal quotients of rp.hpp are correctly rounded for normal operands.)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as O
from pyclaw_amd import _lib as L

orc = O.COracle()
tot = bad = 0
worst = 0.0
for seed in range(30):
    rng = np.random.default_rng(seed)
    mx, my = int(rng.integers(30, 200)), int(rng.integers(30, 200))
    mbc = 2
    shape = (mx + 4, my + 4)
    rho = 0.5 + rng.random(shape)
    p = 0.3 + rng.random(shape)
    scale_u = 10.0 ** rng.uniform(-323, -295, shape)
    scale_v = 10.0 ** rng.uniform(-323, -295, shape)
    mode = seed % 3
    u = (rng.random(shape) - 0.5) * (scale_u if mode != 1 else 1.0)
    v = (rng.random(shape) - 0.5) * (scale_v if mode != 2 else 1.0)
    q0 = np.empty((5,) + shape, order="F")
    q0[0] = rho
    q0[1] = rho * u
    q0[2] = rho * v
    q0[3] = p / 0.4 + 0.5 * rho * (u * u + v * v)
    q0[4] = rng.random(shape) * 10.0 ** rng.uniform(-323, -300, shape)
    par = np.array([1.4, 0.4])
    mth = np.array([4, 4, 4, 4, 2], dtype=np.int32)
    method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
    dx, dy, dt = 1.0 / mx, 0.7 / my, 0.04 / max(mx, my)
    for ids in (1, 2):
        ref = q0.copy("F")
        _, cfl_ref = orc.step2ds(O.RP_EULER5_2D, par, max(mx, my), mbc, mx, my, q0.copy("F"), ref, None, dx, dy, dt,
                                 method, mth, ids)
        out = q0.copy("F")
        cfl = C.c_double()
        L.check(L.lib().pcl_step2ds(O.RP_EULER5_2D, L.d(par), 0, 5, 5, 0, mbc, mx, my, L.d(q0), L.d(out), None, dx, dy,
                                    dt, L.i(method), L.i(mth), C.cast(C.byref(cfl), L.dp), ids))
        n = int((out != ref).sum())
        tot += out.size
        bad += n
        if n:
            d = np.abs(out - ref)
            worst = max(worst, float(d.max()))
            k = np.unravel_index(d.argmax(), d.shape)
            print("seed %d ids %d: %d of %d differ, max abs %g at %s (ref %r out %r) cfl equal %s"
                  % (seed, ids, n, out.size, d.max(), k, ref[k], out[k], cfl.value == cfl_ref))
print("total %d values, %d differ, worst abs diff %g" % (tot, bad, worst))
