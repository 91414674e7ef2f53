"""
The reference's regression problems written against the pyclaw_amd surface, the way the
reference scripts are written against pyclaw (test/euler/2d/shockbubble.py,
test/acoustics/2d/homogeneous/acoustics.py, test/acoustics/1d/homogeneous/acoustics.py,
apps/advection/1d/constant/advection.py).  Shared by the GPU tests, smoke() and bench.py.
"""
import numpy as np

gamma = 1.4
gamma1 = gamma - 1.


def shock_state(pinf=5.):
    rinf = (gamma1 + pinf * (gamma + 1.)) / ((gamma + 1.) + gamma1 * pinf)
    vinf = 1. / np.sqrt(gamma) * (pinf - 1.) / np.sqrt(0.5 * ((gamma + 1.) / gamma) * pinf + 0.5 * gamma1 / gamma)
    einf = 0.5 * rinf * vinf ** 2 + pinf / gamma1
    return rinf, vinf, einf


def sb_qinit(state, x0=0.5, y0=0., r0=0.2, rhoin=0.1, pinf=5.):
    grid = state.grid
    rhoout = 1.
    pout = 1.
    pin = 1.
    x = grid.x.center
    y = grid.y.center
    Y, X = np.meshgrid(y, x)
    r = np.sqrt((X - x0) ** 2 + (Y - y0) ** 2)
    state.q[0, :, :] = rhoin * (r <= r0) + rhoout * (r > r0)
    state.q[1, :, :] = 0.
    state.q[2, :, :] = 0.
    state.q[3, :, :] = (pin * (r <= r0) + pout * (r > r0)) / gamma1
    state.q[4, :, :] = 1. * (r <= r0)


def sb_auxinit(state):
    y = state.grid.y.center
    for j, ycoord in enumerate(y):
        state.aux[0, :, j] = ycoord


def shockbc(state, dim, t, qbc, mbc):
    if dim.nstart == 0:
        rinf, vinf, einf = shock_state()
        for i in range(mbc):
            qbc[0, i, ...] = rinf
            qbc[1, i, ...] = rinf * vinf
            qbc[2, i, ...] = 0.
            qbc[3, i, ...] = einf
            qbc[4, i, ...] = 0.


def euler_rad_src(solver, state, dt):
    dt2 = dt / 2.
    ndim = 2
    aux = state.aux
    q = state.q
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


def dq_euler_radial(solver, state, dt):
    """apps/euler/2d/shockbubble/shockbubble.py:95-122 (dq_Euler_radial): the SharpClaw form of the source."""
    ndim = 2
    q = state.q
    rad = state.aux[0, :, :]
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


def shockbubble(pyclaw, mx=160, my=40, tfinal=0.2, device_callbacks=False, with_src=True,
                dim_split=True, order_trans=2, dt_initial=0.005, nout=1, run=True, math='exact',
                solver_type='classic', time_integrator='SSP104'):
    """test/euler/2d/shockbubble.py:97-166; solver_type='sharpclaw' as in apps/euler/2d/shockbubble/shockbubble.py:173-176
    (dq_src in place of step_src, WENO5).  device_callbacks=True swaps the Python custom-BC / source callbacks for the
    built-in device versions (same arithmetic)."""
    x = pyclaw.Dimension('x', 0.0, 2.0, mx)
    y = pyclaw.Dimension('y', 0.0, 0.5, my)
    grid = pyclaw.Grid([x, y])
    meqn = 5
    maux = 1
    state = pyclaw.State(grid, meqn, maux)
    state.aux_global['gamma'] = gamma
    state.aux_global['gamma1'] = gamma1
    sb_qinit(state)
    sb_auxinit(state)
    initial_solution = pyclaw.Solution(state)

    sharp = solver_type == 'sharpclaw'
    solver = pyclaw.SharpClawSolver2D() if sharp else pyclaw.ClawSolver2D()
    solver.math = math
    solver.rp = pyclaw.riemann.rp_euler_5wave_2d
    solver.mwaves = 5
    solver.dt_initial = dt_initial
    if sharp:
        solver.weno_order = 5
        solver.lim_type = 2
        solver.time_integrator = time_integrator
    else:
        solver.cfl_max = 0.5
        solver.cfl_desired = 0.45
        solver.limiters = [4, 4, 4, 4, 2]
        solver.dim_split = dim_split
        solver.order_trans = order_trans
        solver.src_split = 1
    if device_callbacks:
        rinf, vinf, einf = shock_state()
        solver.user_bc_lower = pyclaw.ConstantStateBC([rinf, rinf * vinf, 0., einf, 0.])
        src = (pyclaw.EulerRadialDqSource if sharp else pyclaw.EulerRadialSource)(gamma1, 2)
    else:
        solver.user_bc_lower = shockbc
        src = dq_euler_radial if sharp else euler_rad_src
    if sharp:
        solver.dq_src = src if with_src else None
    else:
        solver.step_src = src if with_src else None
    solver.bc_lower[0] = pyclaw.BC.custom
    solver.bc_upper[0] = pyclaw.BC.outflow
    solver.bc_lower[1] = pyclaw.BC.reflecting
    solver.bc_upper[1] = pyclaw.BC.outflow
    solver.aux_bc_lower[0] = pyclaw.BC.outflow
    solver.aux_bc_upper[0] = pyclaw.BC.outflow
    solver.aux_bc_lower[1] = pyclaw.BC.outflow
    solver.aux_bc_upper[1] = pyclaw.BC.outflow

    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.tfinal = tfinal
    claw.solution = initial_solution
    claw.solver = solver
    claw.nout = nout
    if not run:
        return claw
    claw.run()
    return claw


def ac2d_qinit(state, width=0.2):
    grid = state.grid
    x = grid.x.center
    y = grid.y.center
    Y, X = np.meshgrid(y, x)
    r = np.sqrt(X ** 2 + Y ** 2)
    state.q[0, :, :] = (np.abs(r - 0.5) <= width) * (1. + np.cos(np.pi * (r - 0.5) / width))
    state.q[1, :, :] = 0.
    state.q[2, :, :] = 0.


def acoustics2D(pyclaw, mx=100, my=100, tfinal=0.12, nout=10, dim_split=1, run=True, math='exact',
                solver_type='classic', lim_type=2, time_integrator='SSP104', weno_order=5):
    """test/acoustics/2d/homogeneous/acoustics.py:19-86 (classic or sharpclaw)."""
    if solver_type == 'classic':
        solver = pyclaw.ClawSolver2D()
    else:
        solver = pyclaw.SharpClawSolver2D()
        solver.lim_type = lim_type
        solver.time_integrator = time_integrator
        solver.weno_order = weno_order
    solver.math = math
    solver.rp = pyclaw.riemann.rp_acoustics_2d
    solver.cfl_max = 0.5
    solver.cfl_desired = 0.45
    solver.mwaves = 2
    solver.dim_split = dim_split
    solver.limiters = [4] * solver.mwaves
    solver.bc_lower[0] = pyclaw.BC.outflow
    solver.bc_upper[0] = pyclaw.BC.outflow
    solver.bc_lower[1] = pyclaw.BC.outflow
    solver.bc_upper[1] = pyclaw.BC.outflow
    x = pyclaw.Dimension('x', -1.0, 1.0, mx)
    y = pyclaw.Dimension('y', -1.0, 1.0, my)
    grid = pyclaw.Grid([x, y])
    state = pyclaw.State(grid, 3)
    rho = 1.0
    bulk = 4.0
    cc = np.sqrt(bulk / rho)
    zz = rho * cc
    state.aux_global['rho'] = rho
    state.aux_global['bulk'] = bulk
    state.aux_global['zz'] = zz
    state.aux_global['cc'] = cc
    ac2d_qinit(state)
    initial_solution = pyclaw.Solution(state)
    solver.dt_initial = np.min(grid.d) / state.aux_global['cc'] * solver.cfl_desired
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.tfinal = tfinal
    claw.solution = initial_solution
    claw.solver = solver
    claw.nout = nout
    if not run:
        return claw
    claw.run()
    return claw


def acoustics1D(pyclaw, mx=100, solver_type='classic', lim_type=2, time_integrator='SSP104', weno_order=5,
                math='exact', char_decomp=0):
    """test/acoustics/1d/homogeneous/acoustics.py: returns the one-period L1 error."""
    if solver_type == 'classic':
        solver = pyclaw.ClawSolver1D()
    else:
        solver = pyclaw.SharpClawSolver1D()
        solver.lim_type = lim_type
        solver.time_integrator = time_integrator
        solver.weno_order = weno_order
        solver.char_decomp = char_decomp
    solver.math = math
    solver.rp = pyclaw.riemann.rp_acoustics_1d
    x = pyclaw.Dimension('x', 0.0, 1.0, mx)
    grid = pyclaw.Grid(x)
    state = pyclaw.State(grid, 2)
    rho = 1.0
    bulk = 1.0
    state.aux_global['rho'] = rho
    state.aux_global['bulk'] = bulk
    state.aux_global['zz'] = np.sqrt(rho * bulk)
    state.aux_global['cc'] = np.sqrt(rho / bulk)
    xc = grid.x.center
    beta = 100
    gam = 0
    x0 = 0.75
    state.q[0, :] = np.exp(-beta * (xc - x0) ** 2) * np.cos(gam * (xc - x0))
    state.q[1, :] = 0.
    init_solution = pyclaw.Solution(state)
    solver.mwaves = 2
    solver.limiters = [4] * solver.mwaves
    solver.dt_initial = grid.d[0] / state.aux_global['cc'] * 0.1
    solver.bc_lower[0] = pyclaw.BC.periodic
    solver.bc_upper[0] = pyclaw.BC.periodic
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.nout = 5
    claw.tfinal = 1.0
    claw.solution = init_solution
    claw.solver = solver
    claw.run()
    q0 = claw.frames[0].state.q.reshape([-1])
    qfinal = claw.frames[claw.nout].state.q.reshape([-1])
    dx = claw.frames[0].grid.d[0]
    return dx * np.sum(np.abs(qfinal - q0)), claw


def advection1D(pyclaw, mx=1000, tfinal=1.0, nout=10):
    """apps/advection/1d/constant/advection.py:3-44 at the C1 size (1000 cells)."""
    solver = pyclaw.ClawSolver1D()
    solver.rp = pyclaw.riemann.rp_advection_1d
    solver.mwaves = 1
    solver.bc_lower[0] = 2
    solver.bc_upper[0] = 2
    x = pyclaw.Dimension('x', 0.0, 1.0, mx)
    grid = pyclaw.Grid(x)
    state = pyclaw.State(grid, 1)
    state.aux_global['u'] = 1.
    xc = grid.x.center
    beta = 100
    gam = 0
    x0 = 0.75
    state.q[0, :] = np.exp(-beta * (xc - x0) ** 2) * np.cos(gam * (xc - x0))
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal = tfinal
    claw.nout = nout
    claw.run()
    return claw


def acoustics3D(pyclaw, test='hom', mx=None, my=None, mz=None, run=True, math='exact', tfinal=2.0, nout=10):
    """test/acoustics/3d/acoustics.py:6-96.  'hom': dim-split 256x4x4, all periodic (the reference gates
    final_difference = 0.00286 +- 1e-4, test/test_examples.py:481-488); 'het' needs the unsplit step3."""
    solver = pyclaw.ClawSolver3D()
    solver.math = math
    solver.rp = pyclaw.riemann.rp_vc_acoustics_3d
    for k in range(3):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.periodic
        solver.aux_bc_lower[k] = solver.aux_bc_upper[k] = pyclaw.BC.periodic
    if test == 'hom':
        solver.dim_split = True
        mx, my, mz = mx or 256, my or 4, mz or 4
        zr = cr = 1.0
    else:
        solver.dim_split = False
        for k in range(3):
            solver.bc_lower[k] = pyclaw.BC.reflecting
            solver.aux_bc_lower[k] = pyclaw.BC.reflecting
        mx, my, mz = mx or 30, my or 30, mz or 30
        zr = cr = 2.0
    solver.mwaves = 2
    solver.limiters = pyclaw.limiters.tvd.MC
    x = pyclaw.Dimension('x', -1.0, 1.0, mx)
    y = pyclaw.Dimension('y', -1.0, 1.0, my)
    z = pyclaw.Dimension('z', -1.0, 1.0, mz)
    grid = pyclaw.Grid([x, y, z])
    state = pyclaw.State(grid, 4, 2)
    zl = cl = 1.0
    grid.compute_c_center()
    X, Y, Z = grid._c_center
    state.aux[0, :, :, :] = zl * (X < 0.) + zr * (X >= 0.)
    state.aux[1, :, :, :] = cl * (X < 0.) + cr * (X >= 0.)
    x0, y0, z0 = -0.5, 0., 0.
    if test == 'hom':
        r = np.sqrt((X - x0) ** 2)
        width = 0.2
        state.q[0, :, :, :] = (np.abs(r) <= width) * (1. + np.cos(np.pi * r / width))
    else:
        r = np.sqrt((X - x0) ** 2 + (Y - y0) ** 2 + (Z - z0) ** 2)
        width = 0.1
        state.q[0, :, :, :] = (np.abs(r - 0.3) <= width) * (1. + np.cos(np.pi * (r - 0.3) / width))
    state.q[1:, :, :, :] = 0.
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal = tfinal
    claw.nout = nout
    if not run:
        return claw
    claw.run()
    return claw


# ---------------------------------------------------------------------------------------------------
# Further set-ups in the style of the reference's apps/ directory (no golden in the reference: the GPU tests check
# them against exact physics or the oracle driver, tests/test_gpu_more_solvers.py)
# ---------------------------------------------------------------------------------------------------
def sod_shock_tube(pyclaw, n=800, tfinal=0.2, solver_type='classic'):
    """apps/euler/1d style: Sod's Riemann problem with rp_euler_1d (Roe + entropy fix)."""
    solver = pyclaw.ClawSolver1D() if solver_type == 'classic' else pyclaw.SharpClawSolver1D()
    solver.rp = pyclaw.riemann.rp_euler_1d
    solver.mwaves = 3
    solver.limiters = [4, 4, 4]
    solver.bc_lower[0] = solver.bc_upper[0] = pyclaw.BC.outflow
    state = pyclaw.State(pyclaw.Grid(pyclaw.Dimension('x', 0.0, 1.0, n)), 3)
    state.aux_global['gamma'] = gamma
    state.aux_global['gamma1'] = gamma1
    xc = state.grid.x.center
    state.q[0] = np.where(xc < 0.5, 1.0, 0.125)
    state.q[1] = 0.0
    state.q[2] = np.where(xc < 0.5, 1.0, 0.1) / gamma1
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = tfinal, 1
    solver.dt_initial = 1e-4
    claw.run()
    return claw


def radial_dam_break(pyclaw, n=200, tfinal=1.0, dim_split=False, solver_type='classic'):
    """apps/shallow/2d style: radial dam break with rp_shallow_2d (Roe + entropy fix, transverse solver)."""
    solver = pyclaw.ClawSolver2D() if solver_type == 'classic' else pyclaw.SharpClawSolver2D()
    solver.rp = pyclaw.riemann.rp_shallow_2d
    solver.mwaves = 3
    solver.limiters = [4, 4, 4]
    if solver_type == 'classic':
        solver.dim_split = dim_split
        solver.order_trans = 2
    for k in range(2):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.outflow
    grid = pyclaw.Grid([pyclaw.Dimension('x', -2.5, 2.5, n), pyclaw.Dimension('y', -2.5, 2.5, n)])
    state = pyclaw.State(grid, 3)
    state.aux_global['g'] = 1.0
    X, Y = grid.c_center
    r = np.sqrt(X ** 2 + Y ** 2)
    state.q[0] = 2.0 * (r <= 0.5) + 1.0 * (r > 0.5)
    state.q[1:] = 0.0
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = tfinal, 1
    solver.dt_initial = 1e-3
    claw.run()
    return claw


def burgers_1d(pyclaw, n=400, tfinal=0.5):
    """apps/burgers/1d style: a step that steepens into a shock moving at the Rankine-Hugoniot speed 1/2."""
    solver = pyclaw.ClawSolver1D()
    solver.rp = pyclaw.riemann.rp_burgers_1d
    solver.mwaves = 1
    solver.limiters = pyclaw.limiters.tvd.vanleer
    solver.bc_lower[0] = solver.bc_upper[0] = pyclaw.BC.outflow
    state = pyclaw.State(pyclaw.Grid(pyclaw.Dimension('x', 0.0, 1.0, n)), 1)
    state.q[0, :] = 1.0 * (state.grid.x.center < 0.25)
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = tfinal, 1
    solver.dt_initial = 0.001
    claw.run()
    return claw


def rotating_flow(pyclaw, n=64, solver_type='classic', tfinal=0.3, run=True):
    """Solid-body-like rotation of a blob: colour equation (rp_vc_advection_2d) with edge velocities from a stream
    function and a capacity function in aux(3) -- the ingredients of apps/advection/2d/annulus on a Cartesian grid --
    periodic in both directions; classic unsplit (order_trans 2, van Leer) or SharpClaw WENO5."""
    if solver_type == "classic":
        solver = pyclaw.ClawSolver2D()
        solver.dim_split = False
        solver.order_trans = 2
        solver.limiters = pyclaw.limiters.tvd.vanleer
        solver.cfl_max, solver.cfl_desired = 1.0, 0.9
    else:
        solver = pyclaw.SharpClawSolver2D()
        solver.lim_type = 2
    solver.rp = pyclaw.riemann.rp_vc_advection_2d
    solver.mwaves = 1
    for k in range(2):
        solver.bc_lower[k] = solver.bc_upper[k] = pyclaw.BC.periodic
        solver.aux_bc_lower[k] = solver.aux_bc_upper[k] = pyclaw.BC.periodic
    grid = pyclaw.Grid([pyclaw.Dimension('x', -1.0, 1.0, n), pyclaw.Dimension('y', -1.0, 1.0, n)])
    state = pyclaw.State(grid, 1, 3)
    state.mcapa = 2
    d = grid.d[0]
    xe, ye = grid.x.edge, grid.y.edge
    psi = lambda x, y: 0.5 * np.pi * (np.cos(np.pi * x / 2) ** 2) * (np.cos(np.pi * y / 2) ** 2)   # stream function
    XE, YE = np.meshgrid(xe, ye, indexing="ij")
    P = psi(XE, YE)
    state.aux[0] = (P[:-1, 1:] - P[:-1, :-1]) / d          # u at the left edge   =  d(psi)/dy
    state.aux[1] = -(P[1:, :-1] - P[:-1, :-1]) / d         # v at the bottom edge = -d(psi)/dx
    X, Y = grid.c_center
    state.aux[2] = 1.0 + 0.2 * np.sin(np.pi * X) * np.sin(np.pi * Y)    # capacity
    state.q[0] = np.exp(-40.0 * ((X - 0.3) ** 2 + Y ** 2))
    claw = pyclaw.Controller()
    claw.keep_copy = True
    claw.solution = pyclaw.Solution(state)
    claw.solver = solver
    claw.tfinal, claw.nout = tfinal, 1
    solver.dt_initial = 0.005
    if run:
        claw.run()
    return claw
