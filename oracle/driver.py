"""
oracle/driver.py -- Python-3 restatement of the reference's *driver* layer, used to
replay the reference's own regression scripts against an oracle back end.
TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

The reference Python layer is 2011 Python 2 and cannot be imported here (SURVEY 8c:
ordinary SyntaxErrors, no permission denial), so this file follows it by reading:

  evolve_to_time      src/pyclaw/solver.py:602-717
  apply_q_bcs / qbc_* src/pyclaw/solver.py:315-452
  ClawSolver.step     src/pyclaw/clawpack.py:114-165
  set_method          src/pyclaw/clawpack.py:192-212
  step_hyperbolic 1-D src/pyclaw/clawpack.py:299-323,405-406
  step_hyperbolic 2-D src/pyclaw/clawpack.py:510-555
  Controller.run      src/pyclaw/controller.py:195-303 (dt_initial once, linspace out times)
  Dimension.d/center  src/pyclaw/grid.py:54-87

It is deliberately procedural and independent of pyclaw_amd/ so that the product's
host logic is checked against a second, separately written statement of the same
semantics.  It is pinned by the reference goldens (tests/test_oracle_golden.py).
"""
import numpy as np

CUSTOM, OUTFLOW, PERIODIC, REFLECTING = 0, 1, 2, 3


def centers(lower, upper, n):
    """grid.py:79-87: center[i] = lower + (i+0.5)*d with d=(upper-lower)/float(n)."""
    d = (upper - lower) / float(n)
    c = np.empty(n)
    for i in range(n):
        c[i] = lower + (i + 0.5) * d
    return c


def fill_ghosts(qbc, mbc, bc_lower, bc_upper, user_lower=None, user_upper=None, t=0.0,
                is_aux=False):
    """solver.py:354-381 with qbc_lower/qbc_upper :384-452 (aux twins :526-596)."""
    ndim = qbc.ndim - 1
    for idim in range(ndim):
        # lower
        bc = bc_lower[idim]
        if bc == CUSTOM:
            user_lower(idim, t, qbc, mbc)
        else:
            v = np.rollaxis(qbc, idim + 1, 1)
            if bc == OUTFLOW:
                for i in range(mbc):
                    v[:, i, ...] = v[:, mbc, ...]
            elif bc == PERIODIC:
                v[:, :mbc, ...] = v[:, -2 * mbc:-mbc, ...]
            elif bc == REFLECTING:
                for i in range(mbc):
                    v[:, i, ...] = v[:, 2 * mbc - 1 - i, ...]
                    if not is_aux:
                        v[idim + 1, i, ...] = -v[idim + 1, 2 * mbc - 1 - i, ...]
            else:
                raise NotImplementedError
        # upper
        bc = bc_upper[idim]
        if bc == CUSTOM:
            user_upper(idim, t, qbc, mbc)
        else:
            v = np.rollaxis(qbc, idim + 1, 1)
            if bc == OUTFLOW:
                for i in range(mbc):
                    v[:, -i - 1, ...] = v[:, -mbc - 1, ...]
            elif bc == PERIODIC:
                v[:, -mbc:, ...] = v[:, mbc:2 * mbc, ...]
            elif bc == REFLECTING:
                for i in range(mbc):
                    v[:, -i - 1, ...] = v[:, -2 * mbc + i, ...]
                    if not is_aux:
                        v[idim + 1, -i - 1, ...] = -v[idim + 1, -2 * mbc + i, ...]
            else:
                raise NotImplementedError


class Problem:
    """Plain record of what the reference spreads over Grid/State/Solver attributes."""

    def __init__(self, **kw):
        self.mbc = 2
        self.order = 2
        self.dim_split = True
        self.order_trans = 1
        self.src_split = 1
        self.fwave = False
        self.step_src = None          # f(problem, q_interior_view, aux, dt)
        self.dq_src = None            # SharpClaw: f(problem, q_stage_interior, aux, dt) -> increment (sharpclaw.py:232-235)
        self.cfl_max = 1.0
        self.cfl_desired = 0.9
        self.dt_initial = 0.1
        self.dt_max = 1e99
        self.max_steps = 1000
        self.dt_variable = True
        self.mcapa = -1
        self.aux = None
        self.user_bc_lower = None
        self.user_bc_upper = None
        self.aux_bc_lower = None
        self.aux_bc_upper = None
        self.user_aux_bc_lower = None      # f(idim, t, auxbc, mbc), solver.py:526-596
        self.user_aux_bc_upper = None
        self.qcor = False                  # app-local step2qcor.f in place of step2.f (shallow water on the sphere)
        self.solver_type = 'classic'
        self.lim_type = 2
        self.weno_order = 5
        self.time_integrator = 'SSP104'
        self.__dict__.update(kw)
        if self.solver_type == 'sharpclaw' and 'mbc' not in kw:
            self.mbc = (self.weno_order + 1) // 2           # sharpclaw.py:479
        self.ndim = self.q.ndim - 1
        self.t = 0.0
        self.dt = self.dt_initial
        # solver.py:172: CFL(_default_attr_values['cfl_desired']) -- the CLASS default (0.9 for
        # ClawSolver, clawpack.py:108; 2.45 for SharpClawSolver, sharpclaw.py:144), not the user's
        # cfl_desired; it seeds status['cflmax'].
        self.cfl = 2.45 if self.solver_type == 'sharpclaw' else 0.9
        self.nrejected = 0
        self.dt_history = []


def setup(p):
    """ClawSolver.setup (clawpack.py:214-238): mthlim, method[7], qbc/auxbc."""
    mth = p.limiters if isinstance(p.limiters, list) else [p.limiters]
    if len(mth) == 1:
        mth = mth * p.mwaves
    assert len(mth) == p.mwaves
    p.mthlim = np.array(mth, dtype=np.int32)
    maux = 0 if p.aux is None else p.aux.shape[0]
    method = np.zeros(7, dtype=np.int32)
    method[0] = int(p.dt_variable)
    method[1] = p.order
    method[2] = 0 if p.ndim == 1 else (-1 if p.dim_split else p.order_trans)
    method[5] = p.mcapa + 1
    method[6] = maux
    p.method = method
    meqn = p.q.shape[0]
    ng = p.q.shape[1:]
    p.qbc = np.zeros((meqn,) + tuple(n + 2 * p.mbc for n in ng), order="F")
    if maux > 0:
        p.auxbc = np.empty((maux,) + tuple(n + 2 * p.mbc for n in ng), order="F")
        inner = (slice(None),) + (slice(p.mbc, -p.mbc),) * p.ndim
        p.auxbc[inner] = p.aux
        fill_ghosts(p.auxbc, p.mbc, p.aux_bc_lower, p.aux_bc_upper, p.user_aux_bc_lower, p.user_aux_bc_upper,
                    is_aux=True)
    else:
        p.auxbc = None


def step_hyperbolic(p, backend):
    mbc = p.mbc
    inner = (slice(None),) + (slice(mbc, -mbc),) * p.ndim
    p.qbc[inner] = p.q                                      # get_qbc_from_q
    fill_ghosts(p.qbc, mbc, p.bc_lower, p.bc_upper, p.user_bc_lower, p.user_bc_upper, p.t)
    if p.ndim == 1:
        mx = p.q.shape[1]
        _, cfl = backend.step1(p.rp, p.rp_params, mbc, mx, p.qbc, p.auxbc, p.d[0], p.dt,
                               p.method, p.mthlim, fwave=p.fwave)
    elif p.ndim == 3:                                       # clawpack.py:650-699 (dim_split only)
        mx, my, mz = p.q.shape[1:]
        maxm = max(mx, my, mz)
        qnew = p.qbc
        qold = qnew.copy("F")
        dx, dy, dz = p.d
        if not p.dim_split:                                 # clawpack.py:690-696: classic3.step3
            _, cfl = backend.step3(p.rp, maxm, mbc, mx, my, mz, qold, qnew, p.auxbc, dx, dy, dz, p.dt, p.method, p.mthlim)
            p.cfl = cfl
            p.q = p.qbc[inner]
            return
        cfl = 0.0
        for idir, qo in ((1, qold), (2, qnew), (3, qnew)):
            _, c1 = backend.step3ds(p.rp, maxm, mbc, mx, my, mz, qo, qnew, p.auxbc, dx, dy, dz, p.dt,
                                    p.method, p.mthlim, idir)
            cfl = max(cfl, c1)
    else:
        mx, my = p.q.shape[1:]
        maxm = max(mx, my)
        qnew = p.qbc
        qold = qnew.copy("F")
        dx, dy = p.d
        if p.dim_split:
            _, cfl_x = backend.step2ds(p.rp, p.rp_params, maxm, mbc, mx, my, qold, qnew, p.auxbc,
                                       dx, dy, p.dt, p.method, p.mthlim, 1, fwave=p.fwave)
            _, cfl_y = backend.step2ds(p.rp, p.rp_params, maxm, mbc, mx, my, qnew, qnew, p.auxbc,
                                       dx, dy, p.dt, p.method, p.mthlim, 2, fwave=p.fwave)
            cfl = max(cfl_x, cfl_y)
        else:
            if p.qcor:
                backend.set_qcor(True)
            try:
                _, cfl = backend.step2(p.rp, p.rp_params, maxm, mbc, mx, my, qold, qnew, p.auxbc,
                                       dx, dy, p.dt, p.method, p.mthlim, fwave=p.fwave)
            finally:
                if p.qcor:
                    backend.set_qcor(False)
    p.cfl = cfl
    p.q = p.qbc[inner]                                      # set_q_from_qbc: a VIEW


def step(p, backend):
    """clawpack.py:142-165"""
    if p.src_split == 2 and p.step_src is not None:
        p.step_src(p, p.q, p.aux, p.dt / 2.0)
    step_hyperbolic(p, backend)
    if p.cfl >= p.cfl_max:
        return False
    if p.step_src is not None:
        if p.src_split == 2:
            p.step_src(p, p.q, p.aux, p.dt / 2.0)
        if p.src_split == 1:
            p.step_src(p, p.q, p.aux, p.dt)
    return True


def evolve_to_time(p, backend, tend):
    """solver.py:602-717 (tend given)."""
    tstart = p.t
    status = {"cflmax": p.cfl, "dtmin": p.dt, "dtmax": p.dt, "numsteps": 0}
    max_steps = p.max_steps
    if not p.dt_variable:
        max_steps = int((tend - tstart + 1e-10) / p.dt)
        if abs(max_steps * p.dt - (tend - tstart)) > 1e-5 * (tend - tstart):
            raise Exception("dt does not divide (tend-tstart) and dt is fixed!")
    if p.dt_variable and p.cfl_desired > p.cfl_max:
        raise Exception("Variable time-stepping and desired CFL > maximum CFL")
    if tend <= tstart:
        max_steps = 0
    for n in range(max_steps):
        if p.t + p.dt > tend and tstart < tend:
            p.dt = tend - p.t
        if p.dt_variable:
            q_backup = p.q.copy("F")
            told = p.t
        if getattr(p, 'solver_type', 'classic') == 'sharpclaw':
            sharp_step(p, backend)
        else:
            step(p, backend)
        cfl = p.cfl
        p.dt_history.append((p.dt, cfl))
        if cfl <= p.cfl_max:
            status["cflmax"] = max(cfl, status["cflmax"])
            if p.dt_variable:
                p.t += p.dt
            else:
                p.t = tstart + (n + 1) * p.dt
            status["numsteps"] += 1
            if p.t >= tend:
                break
        else:
            if p.dt_variable:
                p.q = q_backup
                p.t = told
                p.nrejected += 1
            else:
                status["cflmax"] = max(cfl, status["cflmax"])
                raise Exception("CFL too large, giving up!")
        if p.dt_variable:
            if cfl > 0.0:
                p.dt = min(p.dt_max, p.dt * p.cfl_desired / cfl)
                status["dtmin"] = min(p.dt, status["dtmin"])
                status["dtmax"] = max(p.dt, status["dtmax"])
            else:
                p.dt = p.dt_max
    if p.dt_variable and p.t < tend and status["numsteps"] == max_steps:
        raise Exception("Maximum number of timesteps have been taken")
    return status


# ---------------------------------------------------------------------------------
# SharpClaw: method-of-lines step (src/pyclaw/sharpclaw.py:152-237, 515-563)
# ---------------------------------------------------------------------------------
class CFLError(Exception):
    pass


def sharp_dq(p, backend, q, t):
    """SharpClawSolver.dq -> dq_hyperbolic on an interior array q (a stage register)."""
    mbc = p.mbc
    inner = (slice(None),) + (slice(mbc, -mbc),) * p.ndim
    p.qbc[inner] = q
    fill_ghosts(p.qbc, mbc, p.bc_lower, p.bc_upper, p.user_bc_lower, p.user_bc_upper, t)
    if hasattr(backend, "set_weno_order"):
        backend.set_weno_order(p.weno_order)               # clawparams.weno_order (sharpclaw.py:263)
    if hasattr(backend, "set_char_decomp"):                # clawparams.char_decomp / mthlim (sharpclaw.py:262,268)
        backend.set_char_decomp(getattr(p, "char_decomp", 0))
        if getattr(p, "char_decomp", 0) == 1 or p.lim_type == 1:
            backend.set_sharp_mthlim(p.mthlim)
    if p.ndim == 1:
        dq, cfl = backend.sharp_flux1(p.rp, p.rp_params, p.lim_type, p.mwaves, p.mcapa + 1, mbc, q.shape[1],
                                      p.qbc, p.auxbc, p.d[0], p.dt)
    else:
        mx, my = q.shape[1:]
        dq, cfl = backend.sharp_flux2(p.rp, p.rp_params, p.lim_type, p.mwaves, p.mcapa + 1, mbc, mx, my,
                                      p.qbc, p.auxbc, p.d[0], p.d[1], p.dt)
    p.cfl = cfl                                            # CFL.update_global_max overwrites
    if cfl > p.cfl_max:
        raise CFLError()
    deltaq = dq[inner]
    if p.dq_src is not None:
        deltaq = deltaq + p.dq_src(p, q, p.aux, p.dt)      # deltaq += dq_src(solver, state, dt)
    return deltaq


def sharp_step(p, backend):
    """sharpclaw.py:152-210; q is rebound (not updated in place) exactly like the reference."""
    try:
        q, t, dt = p.q, p.t, p.dt
        if p.time_integrator == 'Euler':
            p.q = q + sharp_dq(p, backend, q, t)
        elif p.time_integrator == 'SSP33':
            s = q + sharp_dq(p, backend, q, t)
            s = 0.75 * q + 0.25 * (s + sharp_dq(p, backend, s, t + dt))
            p.q = 1. / 3. * q + 2. / 3. * (s + sharp_dq(p, backend, s, t + 0.5 * dt))
        elif p.time_integrator == 'SSP104':
            s1 = q + sharp_dq(p, backend, q, t) / 6.
            s1t = t + dt / 6.
            for i in range(4):
                s1 = s1 + sharp_dq(p, backend, s1, s1t) / 6.
                s1t = s1t + dt / 6.
            s2 = q / 25. + 9. / 25 * s1
            s1 = 15. * s2 - 5. * s1
            s1t = t + dt / 3.
            for i in range(4):
                s1 = s1 + sharp_dq(p, backend, s1, s1t) / 6.
                s1t = s1t + dt / 6.
            p.q = s2 + 0.6 * s1 + 0.1 * sharp_dq(p, backend, s1, s1t)
        else:
            raise Exception('Unrecognized time integrator')
    except CFLError:
        return False


def run(p, backend, tfinal, nout=1):
    """Controller.run, outstyle 1 (controller.py:220-231,266-268)."""
    setup(p)
    p.dt = p.dt_initial
    out_times = np.linspace(p.t, tfinal, nout + 1)
    statuses = []
    for t in out_times[1:]:
        statuses.append(evolve_to_time(p, backend, t))
    return statuses


# ---------------------------------------------------------------------------------
# The reference regression problems, restated as data generators
# ---------------------------------------------------------------------------------
GAMMA = 1.4
GAMMA1 = GAMMA - 1.0


def shockbubble_problem(mx=160, my=40, with_src=True, dim_split=True, order_trans=2,
                        dt_initial=0.005, **kw):
    """test/euler/2d/shockbubble.py:9-146 (qinit, auxinit, shockbc, euler_rad_src, solver setup)."""
    from oracle.oracle import RP_EULER5_2D
    gamma, gamma1 = GAMMA, GAMMA1
    x = centers(0.0, 2.0, mx)
    y = centers(0.0, 0.5, my)
    x0, y0, r0, rhoin, pinf = 0.5, 0.0, 0.2, 0.1, 5.0
    Y, X = np.meshgrid(y, x)
    r = np.sqrt((X - x0) ** 2 + (Y - y0) ** 2)
    q = np.empty((5, mx, my), order="F")
    q[0] = rhoin * (r <= r0) + 1.0 * (r > r0)
    q[1] = 0.0
    q[2] = 0.0
    q[3] = (1.0 * (r <= r0) + 1.0 * (r > r0)) / gamma1
    q[4] = 1.0 * (r <= r0)
    aux = np.empty((1, mx, my), order="F")
    for j, yc in enumerate(y):
        aux[0, :, j] = yc

    rinf = (gamma1 + pinf * (gamma + 1.0)) / ((gamma + 1.0) + gamma1 * pinf)
    vinf = 1.0 / np.sqrt(gamma) * (pinf - 1.0) / np.sqrt(0.5 * ((gamma + 1.0) / gamma) * pinf + 0.5 * gamma1 / gamma)
    einf = 0.5 * rinf * vinf ** 2 + pinf / gamma1

    def shockbc(idim, t, qbc, mbc):
        for i in range(mbc):
            qbc[0, i, ...] = rinf
            qbc[1, i, ...] = rinf * vinf
            qbc[2, i, ...] = 0.0
            qbc[3, i, ...] = einf
            qbc[4, i, ...] = 0.0

    def euler_rad_src(p, q, aux, dt):
        dt2 = dt / 2.0
        ndim = 2
        rad = aux[0, :, :]
        rho = q[0, :, :]
        u = q[1, :, :] / rho
        v = q[2, :, :] / rho
        press = gamma1 * (q[3, :, :] - 0.5 * rho * (u ** 2 + v ** 2))
        qstar = np.empty(q.shape)
        qstar[0, :, :] = q[0, :, :] - dt2 * (ndim - 1) / rad * q[2, :, :]
        qstar[1, :, :] = q[1, :, :] - dt2 * (ndim - 1) / rad * rho * u * v
        qstar[2, :, :] = q[2, :, :] - dt2 * (ndim - 1) / rad * rho * v * v
        qstar[3, :, :] = q[3, :, :] - dt2 * (ndim - 1) / rad * v * (q[3, :, :] + press)
        rho = qstar[0, :, :]
        u = qstar[1, :, :] / rho
        v = qstar[2, :, :] / rho
        press = gamma1 * (qstar[3, :, :] - 0.5 * rho * (u ** 2 + v ** 2))
        q[0, :, :] = q[0, :, :] - dt * (ndim - 1) / rad * qstar[2, :, :]
        q[1, :, :] = q[1, :, :] - dt * (ndim - 1) / rad * rho * u * v
        q[2, :, :] = q[2, :, :] - dt * (ndim - 1) / rad * rho * v * v
        q[3, :, :] = q[3, :, :] - dt * (ndim - 1) / rad * v * (qstar[3, :, :] + press)

    def dq_euler_radial(p, q, aux, dt):
        """apps/euler/2d/shockbubble/shockbubble.py:95-122 (dq_Euler_radial)"""
        ndim = 2
        rad = aux[0, :, :]
        rho = q[0, :, :]
        u = q[1, :, :] / rho
        v = q[2, :, :] / rho
        press = gamma1 * (q[3, :, :] - 0.5 * rho * (u ** 2 + v ** 2))
        dq = np.empty(q.shape)
        dq[0, :, :] = -dt * (ndim - 1) / rad * q[2, :, :]
        dq[1, :, :] = -dt * (ndim - 1) / rad * rho * u * v
        dq[2, :, :] = -dt * (ndim - 1) / rad * rho * v * v
        dq[3, :, :] = -dt * (ndim - 1) / rad * v * (q[3, :, :] + press)
        dq[4, :, :] = 0
        return dq

    if kw.get("solver_type") == "sharpclaw":
        kw.setdefault("dq_src", dq_euler_radial if with_src else None)
        with_src = False                                   # the app sets dq_src instead of step_src (shockbubble.py:173-176)
        kw.setdefault("cfl_max", 2.5)                      # the SharpClawSolver defaults (sharpclaw.py:143-144)
        kw.setdefault("cfl_desired", 2.45)
    args = dict(
        q=q, aux=aux, d=(2.0 / float(mx), 0.5 / float(my)),
        rp=RP_EULER5_2D, rp_params=[gamma, gamma1], mwaves=5, limiters=[4, 4, 4, 4, 2],
        cfl_max=0.5, cfl_desired=0.45, dt_initial=dt_initial,
        bc_lower=[CUSTOM, REFLECTING], bc_upper=[OUTFLOW, OUTFLOW], user_bc_lower=shockbc,
        aux_bc_lower=[OUTFLOW, OUTFLOW], aux_bc_upper=[OUTFLOW, OUTFLOW],
        step_src=euler_rad_src if with_src else None, src_split=1,
        dim_split=dim_split, order_trans=order_trans)
    args.update(kw)
    return Problem(**args)


def acoustics2d_problem(mx=100, my=100, dim_split=True, order_trans=2, bcs=None, **kw):
    """test/acoustics/2d/homogeneous/acoustics.py:6-65 (classic variant)."""
    from oracle.oracle import RP_ACOUSTICS_2D
    x = centers(-1.0, 1.0, mx)
    y = centers(-1.0, 1.0, my)
    Y, X = np.meshgrid(y, x)
    r = np.sqrt(X ** 2 + Y ** 2)
    width = 0.2
    q = np.empty((3, mx, my), order="F")
    q[0] = (np.abs(r - 0.5) <= width) * (1.0 + np.cos(np.pi * (r - 0.5) / width))
    q[1] = 0.0
    q[2] = 0.0
    rho, bulk = 1.0, 4.0
    cc = np.sqrt(bulk / rho)
    zz = rho * cc
    d = (2.0 / float(mx), 2.0 / float(my))
    bl, bu = bcs if bcs else ([OUTFLOW, OUTFLOW], [OUTFLOW, OUTFLOW])
    return Problem(
        q=q, d=d, rp=RP_ACOUSTICS_2D, rp_params=[rho, bulk, cc, zz], mwaves=2, limiters=[4] * 2,
        cfl_max=0.5, cfl_desired=0.45, dt_initial=np.min(d) / cc * 0.45,
        bc_lower=bl, bc_upper=bu, dim_split=dim_split, order_trans=order_trans, **kw)


def acoustics1d_problem(mx=100, **kw):
    """test/acoustics/1d/homogeneous/acoustics.py:24-55 (classic Fortran variant; periodic, MC)."""
    from oracle.oracle import RP_ACOUSTICS_1D
    x = centers(0.0, 1.0, mx)
    rho, bulk = 1.0, 1.0
    zz = np.sqrt(rho * bulk)
    cc = np.sqrt(rho / bulk)
    beta, gamma, x0 = 100, 0, 0.75
    q = np.empty((2, mx), order="F")
    q[0, :] = np.exp(-beta * (x - x0) ** 2) * np.cos(gamma * (x - x0))
    q[1, :] = 0.0
    d = (1.0 / float(mx),)
    return Problem(
        q=q, d=d, rp=RP_ACOUSTICS_1D, rp_params=[rho, bulk, cc, zz], mwaves=2, limiters=[4] * 2,
        cfl_max=kw.pop('cfl_max', 1.0), cfl_desired=kw.pop('cfl_desired', 0.9), dt_initial=d[0] / cc * 0.1,
        bc_lower=[PERIODIC], bc_upper=[PERIODIC], **kw)


def advection1d_problem(mx=1000, u=1.0, beta=100.0, x0=0.75):
    """apps/advection/1d/constant/advection.py:28-36 (C1): periodic, Gaussian."""
    from oracle.oracle import RP_ADVECTION_1D
    x = centers(0.0, 1.0, mx)
    q = np.empty((1, mx), order="F")
    q[0, :] = np.exp(-beta * (x - x0) ** 2)
    return Problem(
        q=q, d=(1.0 / float(mx),), rp=RP_ADVECTION_1D, rp_params=[u], mwaves=1, limiters=[1],
        cfl_max=1.0, cfl_desired=0.9, dt_initial=0.1,
        bc_lower=[PERIODIC], bc_upper=[PERIODIC])


def acoustics3d_problem(test='hom', mx=None, my=None, mz=None, **kw):
    """test/acoustics/3d/acoustics.py:6-96 ('hom': dim-split, 256x4x4, all periodic)."""
    if test == 'hom':
        n = (mx or 256, my or 4, mz or 4)
        zr = cr = 1.0
        bc_lower = [PERIODIC] * 3
        dim_split = True
    else:
        n = (mx or 30, my or 30, mz or 30)
        zr = cr = 2.0
        bc_lower = [REFLECTING] * 3
        dim_split = False
    zl = cl = 1.0
    c = [centers(-1.0, 1.0, k) for k in n]
    X, Y, Z = np.meshgrid(c[0], c[1], c[2], indexing="ij")
    aux = np.empty((2,) + n, order="F")
    aux[0] = zl * (X < 0.) + zr * (X >= 0.)
    aux[1] = cl * (X < 0.) + cr * (X >= 0.)
    q = np.zeros((4,) + n, order="F")
    x0, y0, z0 = -0.5, 0., 0.
    if test == 'hom':
        r = np.sqrt((X - x0) ** 2)
        width = 0.2
        q[0] = (np.abs(r) <= width) * (1. + np.cos(np.pi * r / width))
    else:
        r = np.sqrt((X - x0) ** 2 + (Y - y0) ** 2 + (Z - z0) ** 2)
        width = 0.1
        q[0] = (np.abs(r - 0.3) <= width) * (1. + np.cos(np.pi * (r - 0.3) / width))
    from . import oracle as O
    return Problem(q=q, aux=aux, rp=O.RP_VC_ACOUSTICS_3D, rp_params=np.zeros(8), mwaves=2, limiters=4,
                   bc_lower=bc_lower, bc_upper=[PERIODIC] * 3, aux_bc_lower=list(bc_lower),
                   aux_bc_upper=[PERIODIC] * 3, d=tuple(2.0 / k for k in n), dim_split=dim_split,
                   order_trans=22, **kw)


def shallow_sphere_problem(setup_backend, mx=40, my=20, Rsphere=1.0, solver_type='classic'):
    """test/shallow_sphere/shallow_4_Rossby_Haurwitz_wave.py:343-484: 4-Rossby-Haurwitz wave on the sphere, classic
    unsplit (step2qcor) with order_trans=2, MC limiter, capa = aux[0], Strang splitting of the Coriolis source.
    `setup_backend` supplies setaux / qinit / src2 (the C restatement, or the reference's own problem.so)."""
    from oracle.oracle import RP_SHALLOW_SPHERE_2D
    mbc = 2 if solver_type == 'classic' else 3
    xlower, xupper, ylower, yupper = -3.0, 1.0, -1.0, 1.0
    dx, dy = (xupper - xlower) / float(mx), (yupper - ylower) / float(my)
    auxtmp = setup_backend.sphere_setaux(mbc, mx, my, xlower, ylower, dx, dy, Rsphere)     # with ghost cells
    qtmp = setup_backend.sphere_qinit(mbc, mx, my, xlower, ylower, dx, dy, Rsphere)
    aux = np.array(auxtmp[:, mbc:-mbc, mbc:-mbc], order="F")
    q = np.array(qtmp[:, mbc:-mbc, mbc:-mbc], order="F")

    def qbc_lower_y(idim, t, qbc, g):           # :295-303: the ghost rows mirror the first rows, reversed in x
        for j in range(g):
            qbc1D = np.copy(qbc[:, :, 2 * g - 1 - j])
            qbc[:, :, j] = qbc1D[:, ::-1]

    def qbc_upper_y(idim, t, qbc, g):           # :305-313
        for j in range(g):
            qbc1D = np.copy(qbc[:, :, my + g - 1 - j])
            qbc[:, :, my + g + j] = qbc1D[:, ::-1]

    def auxbc_lower_y(idim, t, auxbc, g):       # :316-335: setaux evaluated on the ghost rows
        auxbc[:, :, :g] = auxtmp[:, :, :g]

    def auxbc_upper_y(idim, t, auxbc, g):       # :337-356
        auxbc[:, :, -g:] = auxtmp[:, :, -g:]

    def src(p, qv, auxv, dt):                   # fortran_src_wrapper :24-49 -> src2.f (interior arrays)
        qf = np.array(qv, order="F")
        setup_backend.sphere_src2(qf, np.asfortranarray(auxv), xlower, ylower, dx, dy, dt, Rsphere)
        qv[...] = qf

    if solver_type == 'sharpclaw':
        # BASELINE configs[4]'s synthetic variant (apps/shallow_sphere.py): WENO5 + SSP104 on the same data, SharpClaw
        # defaults (sharpclaw.py:127-146), no source term
        return Problem(
            q=q, aux=aux, d=(dx, dy), rp=RP_SHALLOW_SPHERE_2D, rp_params=[11489.57219, dx, dy], mwaves=3, limiters=[1],
            solver_type='sharpclaw', lim_type=2, mcapa=0, cfl_max=2.5, cfl_desired=2.45,
            bc_lower=[PERIODIC, CUSTOM], bc_upper=[PERIODIC, CUSTOM],
            user_bc_lower=qbc_lower_y, user_bc_upper=qbc_upper_y,
            aux_bc_lower=[PERIODIC, CUSTOM], aux_bc_upper=[PERIODIC, CUSTOM],
            user_aux_bc_lower=auxbc_lower_y, user_aux_bc_upper=auxbc_upper_y)
    return Problem(
        q=q, aux=aux, d=(dx, dy), rp=RP_SHALLOW_SPHERE_2D, rp_params=[11489.57219, dx, dy], mwaves=3, limiters=4,
        dim_split=False, order_trans=2, src_split=2, step_src=src, mcapa=0, qcor=True,
        bc_lower=[PERIODIC, CUSTOM], bc_upper=[PERIODIC, CUSTOM],
        user_bc_lower=qbc_lower_y, user_bc_upper=qbc_upper_y,
        aux_bc_lower=[PERIODIC, CUSTOM], aux_bc_upper=[PERIODIC, CUSTOM],
        user_aux_bc_lower=auxbc_lower_y, user_aux_bc_upper=auxbc_upper_y)
