"""
Worker for tests/test_gpu_multiproc.py: one of N processes that SHARE ONE GPU (PCL_FORCE_DEVICE=0) and run the
product's decomposed device path end to end -- decomposition, device pack / unpack kernels, neighbour tables, the
interior / rim stream choreography, boundary conditions on edge blocks only, the CFL maximum over the blocks -- with
the host-staged wire (PCL_HALO_TRANSPORT=host: the strips travel over pyclaw_amd.parallel's TCP group instead of
RCCL, which refuses two ranks on one device).  Rank 0 gathers the blocks and compares the assembled field with the
SERIAL oracle replay: bit for bit.

  python tests/mp_gpu_worker.py <case>       with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment
"""
import base64
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import pyclaw_amd as pyclaw                  # noqa: E402
from pyclaw_amd import parallel             # noqa: E402
from apps import problems                   # noqa: E402


def run_case(case):
    if case == "sphere_classic":
        # C5's physics on several ranks: blocks cut in y only (the pole boundary reverses whole rows); set-up arrays
        # from the C restatement of the app's setaux.f / qinit.f so that the serial oracle replay starts from the same bits
        from apps import shallow_sphere as S
        from oracle import oracle as O
        co = O.COracle()
        mx, my = 40, 20
        aux = co.sphere_setaux(2, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)
        q0 = co.sphere_qinit(2, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)[:, 2:-2, 2:-2]
        return S.shallow_sphere(pyclaw, mx, my, tfinal=2.0, nout=2, aux_full=aux, q0=q0)
    if case == "sphere_sharpclaw":
        # C5 as BASELINE configures it: SharpClaw WENO5 + SSP104 on the sphere (mbc = 3): every Runge-Kutta stage
        # exchanges its halo and mirrors the pole rows (shallow_4_Rossby_Haurwitz_wave.py:295-313) on the edge blocks
        from apps import shallow_sphere as S
        from oracle import oracle as O
        co = O.COracle()
        mx, my = 48, 24
        aux = co.sphere_setaux(3, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)
        q0 = co.sphere_qinit(3, mx, my, -3.0, -1.0, 4.0 / mx, 2.0 / my)[:, 3:-3, 3:-3]
        return S.shallow_sphere(pyclaw, mx, my, tfinal=0.5, nout=1, aux_full=aux, q0=q0, solver_type='sharpclaw')
    if case == "advection1d":               # 1-D grids are cut too (petclaw/state.py:199-234): W / E strips, periodic wrap
        return problems.advection1D(pyclaw, mx=1000, tfinal=0.3, nout=2)
    if case == "acoustics1d_sharp":         # SharpClaw in 1-D: every Runge-Kutta stage exchanges its 3-cell halo
        return problems.acoustics1D(pyclaw, mx=300, solver_type='sharpclaw')[1]
    if case == "shockbubble_ds":
        claw = problems.shockbubble(pyclaw, mx=160, my=40, tfinal=0.03, device_callbacks=True)
    elif case == "shockbubble_unsplit":
        claw = problems.shockbubble(pyclaw, mx=160, my=40, tfinal=0.03, device_callbacks=True, dim_split=False)
    elif case == "shockbubble_pycb":         # the reference app's own Python callbacks: custom-BC strips + numpy source term
        claw = problems.shockbubble(pyclaw, mx=160, my=40, tfinal=0.03, device_callbacks=False)
    elif case == "acoustics_odd":            # sizes that do not divide: blocks of unequal width (PETSc: remainder first)
        claw = problems.acoustics2D(pyclaw, mx=91, my=83, tfinal=0.06, nout=2, dim_split=0)
    elif case == "acoustics_ds":
        claw = problems.acoustics2D(pyclaw, mx=90, my=80, tfinal=0.06, nout=2)
    elif case == "acoustics_unsplit":
        claw = problems.acoustics2D(pyclaw, mx=90, my=80, tfinal=0.06, nout=2, dim_split=0)
    elif case == "acoustics_ds_mbc3":        # three ghost layers: the outer layer is copied through next to the exchange
        claw = problems.acoustics2D(pyclaw, mx=1000, my=40, tfinal=0.03, nout=1, run=False)
        claw.solver.mbc = 3
        claw.run()
    elif case == "acoustics_sharp":
        claw = problems.acoustics2D(pyclaw, mx=90, my=80, tfinal=0.03, nout=1, solver_type='sharpclaw')
    elif case == "acoustics_sharp9":
        claw = problems.acoustics2D(pyclaw, mx=90, my=80, tfinal=0.03, nout=1, solver_type='sharpclaw', weno_order=9,
                                    time_integrator='SSP33')
    elif case == "acoustics3d_ds":
        claw = problems.acoustics3D(pyclaw, test='hom', mx=40, my=18, mz=14, tfinal=0.3, nout=1)
    elif case == "acoustics3d_unsplit":
        claw = problems.acoustics3D(pyclaw, test='het', mx=24, my=20, mz=16, tfinal=0.3, nout=1)
    elif case.startswith("rotating_"):
        claw = problems.rotating_flow(pyclaw, n=64, solver_type=case.split("_")[1], tfinal=0.15)
    else:
        raise SystemExit("unknown case " + case)
    return claw


def oracle_case(case):
    from oracle import driver as D
    from oracle import oracle as O
    co = O.COracle()
    if case == "sphere_classic":
        p = D.shallow_sphere_problem(co)
        D.run(p, co, 2.0, 2)
        return p.q, p
    if case == "sphere_sharpclaw":
        p = D.shallow_sphere_problem(co, mx=48, my=24, solver_type='sharpclaw')
        D.run(p, co, 0.5, 1)
        return p.q, p
    if case == "advection1d":
        p = D.advection1d_problem(mx=1000)
        D.run(p, co, 0.3, 2)
        return p.q, p
    if case == "acoustics1d_sharp":
        p = D.acoustics1d_problem(mx=300, solver_type='sharpclaw', cfl_max=2.5, cfl_desired=2.45)
        D.run(p, co, 1.0, 5)
        return p.q, p
    if case.startswith("shockbubble"):
        p = D.shockbubble_problem(mx=160, my=40, dim_split=not case.endswith("_unsplit"))
        D.run(p, co, 0.03, 1)
    elif case == "acoustics_sharp":
        p = D.acoustics2d_problem(mx=90, my=80, solver_type='sharpclaw')
        D.run(p, co, 0.03, 1)
    elif case == "acoustics_sharp9":
        p = D.acoustics2d_problem(mx=90, my=80, solver_type='sharpclaw', weno_order=9, time_integrator='SSP33')
        D.run(p, co, 0.03, 1)
    elif case == "acoustics_odd":
        p = D.acoustics2d_problem(mx=91, my=83, dim_split=False, order_trans=1)
        D.run(p, co, 0.06, 2)
    elif case == "acoustics_ds_mbc3":
        p = D.acoustics2d_problem(mx=1000, my=40, dim_split=True, order_trans=1, mbc=3)
        D.run(p, co, 0.03, 1)
    elif case == "acoustics3d_ds":
        p = D.acoustics3d_problem('hom', mx=40, my=18, mz=14)
        D.run(p, co, 0.3, 1)
    elif case == "acoustics3d_unsplit":
        p = D.acoustics3d_problem('het', mx=24, my=20, mz=16)
        D.run(p, co, 0.3, 1)
    elif case.startswith("rotating_"):
        # the serial set-up arrays come from the same app function on an undecomposed grid: build them with numpy here
        st = case.split("_")[1]
        n = 64
        d = 2.0 / n
        xe = -1.0 + d * np.arange(n + 1)
        xc = -1.0 + d * (np.arange(n) + 0.5)
        psi = lambda x, y: 0.5 * np.pi * (np.cos(np.pi * x / 2) ** 2) * (np.cos(np.pi * y / 2) ** 2)
        XE, YE = np.meshgrid(xe, xe, indexing="ij")
        P = psi(XE, YE)
        X, Y = np.meshgrid(xc, xc, indexing="ij")
        aux = np.empty((3, n, n), order="F")
        aux[0] = (P[:-1, 1:] - P[:-1, :-1]) / d
        aux[1] = -(P[1:, :-1] - P[:-1, :-1]) / d
        aux[2] = 1.0 + 0.2 * np.sin(np.pi * X) * np.sin(np.pi * Y)
        q0 = np.empty((1, n, n), order="F")
        q0[0] = np.exp(-40.0 * ((X - 0.3) ** 2 + Y ** 2))
        kw = (dict(solver_type='sharpclaw', lim_type=2, cfl_max=2.5, cfl_desired=2.45) if st == "sharpclaw"
              else dict(cfl_max=1.0, cfl_desired=0.9))
        p = D.Problem(q=q0, aux=aux, rp=O.RP_VC_ADVECTION_2D, rp_params=np.zeros(8), mwaves=1, limiters=3,
                      bc_lower=[D.PERIODIC] * 2, bc_upper=[D.PERIODIC] * 2, aux_bc_lower=[D.PERIODIC] * 2,
                      aux_bc_upper=[D.PERIODIC] * 2, d=(d, d), dim_split=False, order_trans=2, mcapa=2,
                      dt_initial=0.005, **kw)
        D.run(p, co, 0.15, 1)
    else:
        # order_trans: the solver default (trans_inc, clawpack.py:460), which apps/problems.acoustics2D keeps
        p = D.acoustics2d_problem(mx=90, my=80, dim_split=case.endswith("_ds"), order_trans=1)
        D.run(p, co, 0.06, 2)
    return p.q, p


def main():
    case = sys.argv[1]
    if case.startswith("sphere"):
        os.environ["PCL_PROC_GRID"] = "1x%s" % os.environ.get("WORLD_SIZE", "1")
    parallel.init()
    rank, size = parallel.rank(), parallel.world_size()
    claw = run_case(case)
    st = claw.frames[claw.nout].state
    rng = [(d.nstart, d.nend) for d in st.grid.dimensions]
    g = parallel._state["group"]
    blocks = g.allgather({"rank": rank, "rng": rng, "shape": list(st.q.shape),
                          "q": base64.b64encode(np.ascontiguousarray(st.q).tobytes()).decode("ascii"),
                          "steps": int(claw.solver.status["numsteps"])})
    ok = 0
    if rank == 0:
        ref, p = oracle_case(case)
        full = np.full(ref.shape, np.nan)
        for b in blocks:
            q = np.frombuffer(base64.b64decode(b["q"]), dtype=np.float64).reshape(b["shape"])
            idx = (slice(None),) + tuple(slice(a, b_) for a, b_ in b["rng"])
            full[idx] = q
        nb = len(set(tuple(map(tuple, b["rng"])) for b in blocks))
        same = np.array_equal(full, ref)
        if not same:
            bad = np.argwhere(~((full == ref) | (np.isnan(full) & np.isnan(ref))))
            print("mismatches: %d cells; NaN in result %d, in reference %d; first at %s; index ranges %s"
                  % (len(bad), np.isnan(full).sum(), np.isnan(ref).sum(), bad[:6].tolist(),
                     [(int(bad[:, k].min()), int(bad[:, k].max())) for k in range(1, bad.shape[1])]))
        print("case %s: %d ranks, %d distinct blocks, steps %s, max |diff| %g, bit-identical: %s"
              % (case, size, nb, sorted(set(b["steps"] for b in blocks)), np.nanmax(np.abs(full - ref)), same))
        ok = 0 if (same and nb == size) else 3
    parallel.barrier()
    parallel.shutdown()
    sys.exit(ok)


if __name__ == "__main__":
    main()
