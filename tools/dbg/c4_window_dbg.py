"""debug: which cells / components / steps of the c4_periodic window differ from the oracle"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mp_fullsize_worker as W
import test_gpu_c4c5 as T
from oracle import oracle as O
import pyclaw_amd as pyclaw

co = O.COracle()
case = sys.argv[1] if len(sys.argv) > 1 else "c4_periodic"
for (nx, ny) in [(256, 192), (2048, 2048), (8192, 8192)]:
    for steps in (1, 3):
        for src in (True, False):
            W.CASES[case] = (nx, ny, "2x2", steps)
            claw = W.build(case, pyclaw)
            if not src:
                claw.solver.step_src = None
            claw.run()
            q = claw.solution.state.q
            w, pad = 24, 2 * steps + 2
            hx, hy = nx // 2, ny // 2
            bad = 0
            for (i0, j0) in [(hx - 12, hy - 12), (hx - 12, hy // 2), (-12, -12), (hx + 31, hy // 2 + 17)]:
                # replay (optionally without source)
                ii = np.arange(i0 - pad, i0 + w + pad) % nx
                jj = np.arange(j0 - pad, j0 + w + pad) % ny
                from oracle import driver as D
                q0 = W.synth_euler(ii, jj)
                dx, dy = 2.0 / float(nx), (2.0 * ny / nx) / float(ny)
                aux = np.empty((1, len(ii), len(jj)), order="F"); aux[0] = ((jj + 0.5) * dy)[None, :]
                dt = W.fixed_dt(case)
                n = w + 2 * pad
                p = D.shockbubble_problem(mx=n, my=n, q=q0, aux=aux, d=(dx, dy), dim_split=True, order_trans=2,
                                          bc_lower=[D.OUTFLOW] * 2, bc_upper=[D.OUTFLOW] * 2, user_bc_lower=None,
                                          cfl_max=1.0, cfl_desired=0.9, dt_initial=dt, dt_variable=False, with_src=src)
                D.run(p, co, steps * dt, 1)
                ref = p.q[:, pad:-pad, pad:-pad]
                got = T.wrap_take(q, i0, j0, w)
                d = np.argwhere(got != ref)
                if len(d):
                    bad += 1
                    print("  n=%dx%d steps=%d src=%s window (%d,%d): %d cells differ, comps %s, max %.3g, first %s got %r ref %r"
                          % (nx, ny, steps, src, i0, j0, len(d), sorted(set(d[:, 0])), np.abs(got - ref).max(), d[0],
                             got[tuple(d[0])], ref[tuple(d[0])]))
            print("n=%dx%d steps=%d src=%s: %d bad windows; cflmax %r dt %r t %r" % (nx, ny, steps, src, bad,
                  claw.solver.status['cflmax'], claw.solver.dt, claw.solution.t))
