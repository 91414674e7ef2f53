"""
Worker of tests/test_gpu_fused_step.py: a few dimension-split 2-D problems through Controller.run on one GPU; prints one
JSON line with a hash of every final state and the step / Courant-number bookkeeping.  The caller runs it twice --
PCL_TUNE_FUSED_STEP=1 (default: both sweeps of a step in one kernel, classic_fused.hpp) and =0 (x pass + y pass,
classic.hpp) -- and compares the lines: the two forms of step2ds.f must agree bit for bit.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import pyclaw_amd as pyclaw                     # noqa: E402
from apps import problems                       # noqa: E402
import mp_fullsize_worker as W                  # noqa: E402

LAST_FORMS = None


def finish(claw, zero_signs=True):
    st = claw.solution.state
    global LAST_FORMS
    claw.solver.teardown()          # (a second call is harmless) leaves status["step_forms"]
    f = claw.solver.status.get("step_forms")
    if f is not None:
        LAST_FORMS = [f["one_kernel"], f["two_pass"]]
    # zero_signs=False: -0.0 hashed as +0.0.  A wavefront without a jump hands its cells through as they are, the full
    # solve computes q + 0 like the reference -- which turns a -0.0 into +0.0; the two forms of the step cut the grid
    # into different wavefronts, so they may differ in the SIGN of a zero where the state holds negative zeros (DESIGN
    # 4.1: the accepted difference), and in nothing else
    out = {"hash": W.block_hash(st.q if zero_signs else st.q + 0.0), "steps": int(claw.solver.status["numsteps"]),
           "cflmax": repr(float(claw.solver.status["cflmax"])), "dt": repr(float(claw.solver.dt)),
           "finite": bool(np.isfinite(st.q).all())}
    claw.solver.teardown()
    return out


def patchwork(mx, my):
    """constant patches (37 x 9 cells, three states, momenta that are +0 in one patch and -0 in the next): wavefronts
    without a jump that meet the same state again, another undisturbed state, or the same state up to the sign of a
    zero -- what the remembered-state shortcut of the one-kernel step (NoJumpMemo, classic.hpp) has to get right --
    next to patch edges, where the full solve runs"""
    states = np.array([[1.0, 0.0, 0.0, 2.5, 0.0], [1.0, -0.0, -0.0, 2.5, 0.0], [0.5, 0.3, -0.1, 1.7, 1.0],
                       [1.0, 0.0, -0.0, 2.5, 0.0], [2.0, -0.4, 0.2, 6.0, 0.25]])
    i, j = np.meshgrid(np.arange(mx), np.arange(my), indexing='ij')
    k = ((i // 37) * 3 + (j // 9) * 2 + (i // 74)) % len(states)
    return np.moveaxis(states[k], -1, 0).copy()


def synthetic_euler(mx, my, bc, limiters, order=2, src=True, steps=4, init=None, zero_signs=True):
    """the dense synthetic state of the full-size tests on an mx x my grid with the given boundary conditions"""
    x = pyclaw.Dimension('x', 0.0, 2.0, mx)
    y = pyclaw.Dimension('y', 0.0, 2.0 * my / mx, my)
    state = pyclaw.State(pyclaw.Grid([x, y]), 5, 1)
    state.aux_global['gamma'] = W.GAMMA
    state.aux_global['gamma1'] = W.GAMMA1
    state.q[...] = W.synth_euler(np.arange(mx), np.arange(my)) if init is None else init(mx, my)
    problems.sb_auxinit(state)
    solver = pyclaw.ClawSolver2D()
    solver.rp = pyclaw.riemann.rp_euler_5wave_2d
    solver.mwaves = 5
    solver.limiters = limiters
    solver.order = order
    solver.dim_split = True
    if src:
        solver.src_split = 1
        solver.step_src = pyclaw.EulerRadialSource(W.GAMMA1, 2)
    solver.cfl_max, solver.cfl_desired = 1.0, 0.9
    solver.dt_variable = False
    solver.dt_initial = 0.05 / max(mx, my)
    for k in range(2):
        solver.bc_lower[k], solver.bc_upper[k] = bc[2 * k], bc[2 * k + 1]
        solver.aux_bc_lower[k] = solver.aux_bc_upper[k] = pyclaw.BC.outflow
    claw = pyclaw.Controller()
    claw.keep_copy = False
    claw.output_format = None
    claw.tfinal = steps * solver.dt_initial
    claw.nout = 1
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.run()
    return finish(claw, zero_signs)


def main():
    B = pyclaw.BC
    res = {}
    # the app itself: inflow / reflecting / outflow sides, adaptive dt incl. a rejected step, fused source
    for (mx, my) in ((160, 40), (333, 61), (64, 28), (61, 29), (60, 12), (61, 13), (1030, 250)):
        claw = problems.shockbubble(pyclaw, mx=mx, my=my, tfinal=0.02, device_callbacks=True, run=False)
        claw.keep_copy = False
        claw.output_format = None
        claw.run()
        res["shockbubble_%dx%d" % (mx, my)] = finish(claw)
    # every built-in boundary condition on every side, tile-edge sizes (60 owned columns, 12 owned rows per tile -- and
    # the 28 rows of the 32-row tile shape), first order, other limiters, no source
    per, out, ref = B.periodic, B.outflow, B.reflecting
    for tag, (mx, my), bc, lim, order, src in (
            ("periodic_120x56", (120, 56), [per, per, per, per], [4, 4, 4, 4, 2], 2, True),
            ("periodic_121x57", (121, 57), [per, per, per, per], [4, 4, 4, 4, 2], 2, False),
            ("periodic_120x24", (120, 24), [per, per, per, per], [4, 4, 4, 4, 2], 2, True),
            ("periodic_121x25", (121, 25), [per, per, per, per], [4, 4, 4, 4, 2], 2, False),
            ("mixed_59x11", (59, 11), [ref, out, out, ref], [4, 4, 4, 4, 2], 2, True),
            ("mixed_59x27", (59, 27), [out, ref, ref, out], [1, 2, 3, 4, 0], 2, True),
            ("mixed_200x90", (200, 90), [ref, out, per, per], [3, 3, 3, 3, 3], 2, False),
            ("order1_77x33", (77, 33), [out, out, ref, ref], [4, 4, 4, 4, 2], 1, True),
            ("thin_300x5", (300, 5), [per, per, out, out], [4, 4, 4, 4, 2], 2, False),
            ("narrow_3x90", (3, 90), [ref, ref, per, per], [4, 4, 4, 4, 2], 2, False)):
        res[tag] = synthetic_euler(mx, my, bc, lim, order, src)
    # constant patches: the remembered undisturbed state of the one-kernel step (hits, misses, +0 / -0)
    res["patchwork_300x100"] = synthetic_euler(300, 100, [per, per, out, ref], [4, 4, 4, 4, 2], 2, False, steps=6, init=patchwork, zero_signs=False)
    res["patchwork_src_190x61"] = synthetic_euler(190, 61, [out, ref, per, per], [4, 4, 4, 4, 2], 2, True, steps=3, init=patchwork, zero_signs=False)
    res["patchwork_poszero_300x100"] = synthetic_euler(300, 100, [per, per, out, ref], [4, 4, 4, 4, 2], 2, False, steps=6,
                                                       init=lambda mx, my: np.abs(patchwork(mx, my)) * np.array([1, 1, -1, 1, 1.0]).reshape(5, 1, 1) + 0.0)
    # 80 steps: in the default mode (PCL_TUNE_FUSED_STEP=2) the solver's trial steps (64..71 of a window) run both forms
    res["window_200x90"] = synthetic_euler(200, 90, [per, per, out, ref], [4, 4, 4, 4, 2], 2, True, steps=80)
    forms = LAST_FORMS
    # the other aux-free 2-D solvers
    claw = problems.acoustics2D(pyclaw, mx=130, my=75, tfinal=0.05, nout=1, dim_split=1, run=False)
    claw.keep_copy = False
    claw.output_format = None
    claw.run()
    res["acoustics_130x75"] = finish(claw)
    claw = problems.radial_dam_break(pyclaw, n=150, tfinal=0.1, dim_split=True)
    res["dam_break_150"] = finish(claw) if hasattr(claw, "solver") else {"hash": W.block_hash(np.asarray(claw))}
    sys.stderr.write("forms %s\n" % json.dumps(forms))      # not part of the compared line
    print(json.dumps(res))


if __name__ == "__main__":
    main()
