#!/usr/bin/env python
"""
huge_grid_check.py -- spot parity on a grid whose arrays hold more than 2^31 doubles (default 22528^2 Euler:
2.5e9 doubles, 20 GB per buffer): one dim-split step on the GPU, windows of the result -- at random places and in
the far corners, i.e. beyond the 32-bit element offset -- against the CPU oracle on the same cells, bit for bit.
Needs ~45 GB of host memory and ~100 GB of HBM.   python tools/huge_grid_check.py [n]
"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O                      # noqa: E402
from pyclaw_amd import _lib as L                    # noqa: E402
import test_gpu_fullsize as T                       # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 22528
    lib = L.lib()
    co = O.COracle()
    rng = np.random.default_rng(7)
    mth = [4, 4, 4, 4, 2]
    dt = 0.1 / n
    t0 = time.time()
    qbc = T.euler_field(rng, (n + 4, n + 4))
    print("state built: %.1f GB, %.0f s" % (qbc.nbytes / 1e9, time.time() - t0), flush=True)
    cflp = C.c_double()
    h = T.make(L, 2, (n, n), O.RP_EULER5_2D, 5, 5, mth, -1)
    try:
        L.check(lib.pcl_put_q(h, L.d(qbc), 1))
        L.check(lib.pcl_step_hyperbolic(h, dt, C.cast(C.byref(cflp), L.dp)))
        out = np.empty_like(qbc)
        L.check(lib.pcl_get_q(h, L.d(out), 1))
    finally:
        lib.pcl_destroy(h)
    print("stepped: cfl %.6f, %.0f s" % (cflp.value, time.time() - t0), flush=True)
    assert 0 < cflp.value < 1
    method = np.array([1, 2, -1, 0, 0, 0, 0], dtype=np.int32)
    w = 48
    corners = list(T.windows(rng, (n + 4, n + 4), w, 8, 8)) + [(4, 4), (n + 4 - w - 6, 4), (4, n + 4 - w - 6),
                                                                (n + 4 - w - 6, n + 4 - w - 6), (n // 3 - 10, n // 5 - 10)]
    for (i0, j0) in corners:
        blk = np.array(qbc[:, i0 - 2:i0 + w + 2, j0 - 4:j0 + w + 4], order="F")
        ref = blk.copy("F")
        co.step2ds(O.RP_EULER5_2D, [1.4, 0.4], w + 4, 2, w, w + 4, blk, ref, None, 1.0 / n, 1.0 / n, dt, method, mth, 1)
        co.step2ds(O.RP_EULER5_2D, [1.4, 0.4], w + 4, 2, w, w + 4, ref, ref, None, 1.0 / n, 1.0 / n, dt, method, mth, 2)
        same = np.array_equal(out[:, i0:i0 + w, j0:j0 + w], ref[:, 2:-2, 4:-4])
        flat = (4 * (n + 4) * (n + 4) + j0 * (n + 4) + i0)
        print("window at (%d, %d): %s   (element offset of its last plane ~ %.2e)" % (i0, j0, "bit-identical" if same else "DIFFERS", flat))
        assert same
    print("huge grid check passed: n = %d, %d windows, %.0f s" % (n, len(corners), time.time() - t0))


if __name__ == "__main__":
    main()
