"""
Riemann solver registry.

The reference picks the Riemann solver when the app's f2py module is linked
(``apps/*/Makefile``: ``$(RIEMANN)/src/rpn2_*.f``) and hands its scalars over through the
``cparam`` common block filled from ``state.aux_global`` (src/pyclaw/state.py:142-162).
Here a solver is named on the Solver object (``solver.rp = riemann.rp_euler_5wave_2d`` or
the string ``'euler_5wave_2d'``); ``cparam`` lists the aux_global keys in common-block order.
"""


class RiemannSolver(object):
    def __init__(self, name, rp_id, ndim, meqn, mwaves, cparam, has_transverse=False, grid_params=False):
        self.name = name
        self.grid_params = grid_params     # append the grid spacings (the reference app sets comxyt.dxcom/dycom)
        self.id = rp_id
        self.ndim = ndim
        self.meqn = meqn
        self.mwaves = mwaves
        self.cparam = tuple(cparam)
        self.has_transverse = has_transverse

    def params(self, aux_global):
        """cparam values from aux_global; same check as State.set_cparam (state.py:156-160)."""
        if not set(self.cparam) <= set(aux_global.keys()):
            raise Exception("""Some required value(s) in the cparam common 
                                   block in the Riemann solver have not been 
                                   set in aux_global.""")
        return [float(aux_global[k]) for k in self.cparam]

    def all_params(self, state):
        """cparam values, then dx, dy(, dz) for solvers that read common /comxyt/ (set by the reference app:
        apps/shallow-sphere/shallow_4_Rossby_Haurwitz_wave.py:449-451)."""
        p = self.params(state.aux_global)
        if self.grid_params:
            p = p + [float(d) for d in state.grid.d]
        return p

    def __repr__(self):
        return "<RiemannSolver %s>" % self.name


rp_advection_1d = RiemannSolver("advection_1d", 1, 1, 1, 1, ["u"])
rp_acoustics_1d = RiemannSolver("acoustics_1d", 2, 1, 2, 2, ["rho", "bulk", "cc", "zz"])
rp_advection_color_1d = RiemannSolver("advection_color_1d", 6, 1, 1, 1, [])        # aux(1) = velocity at the left edge
rp_burgers_1d = RiemannSolver("burgers_1d", 3, 1, 1, 1, [])
rp_euler_1d = RiemannSolver("euler_1d", 4, 1, 3, 3, ["gamma", "gamma1"])            # rp1_euler_with_efix
rp_shallow_1d = RiemannSolver("shallow_1d", 5, 1, 2, 2, ["g"])                       # rp1_shallow_roe_with_efix
# f-wave solvers (solver.fwave = True, the reference's classic1fw / classic2fw): aux(1)=rho, aux(2)=K, aux(3)=1 for the
# linear stress law, anything else for sigma = exp(K eps) - 1; the p-system's transverse solver reads aux(4) = eps
rp_elasticity_fwave_1d = RiemannSolver("elasticity_fwave_1d", 7, 1, 2, 2, [])
rp_psystem_fwave_2d = RiemannSolver("psystem_fwave_2d", 17, 2, 3, 2, [], True)
rp_advection_2d = RiemannSolver("advection_2d", 12, 2, 1, 1, ["u", "v"], True)
rp_shallow_2d = RiemannSolver("shallow_2d", 13, 2, 3, 3, ["g"], True)             # rpn2/rpt2_shallow_roe_with_efix
rp_vc_acoustics_2d = RiemannSolver("vc_acoustics_2d", 14, 2, 3, 2, [], True)     # aux(1)=Z, aux(2)=c
rp_vc_advection_2d = RiemannSolver("vc_advection_2d", 15, 2, 1, 1, [], True)     # aux(1)=u at left edge, aux(2)=v at bottom edge
rp_acoustics_2d = RiemannSolver("acoustics_2d", 10, 2, 3, 2, ["rho", "bulk", "cc", "zz"], True)
rp_euler_5wave_2d = RiemannSolver("euler_5wave_2d", 11, 2, 5, 5, ["gamma", "gamma1"], True)
# rpn2/rpt2_shallow_sphere (apps/shallow-sphere/Makefile:7): common /sw/ g; the 16 aux components of setaux.f; the
# unsplit step is the app's step2qcor.f
rp_shallow_sphere_2d = RiemannSolver("shallow_sphere_2d", 16, 2, 4, 3, ["g"], True, grid_params=True)

# 3-D acoustics, impedance and sound speed per cell in aux(1), aux(2) (test/acoustics/3d/Makefile:
# rpn3_vc_acoustics.f; the transverse rpt3/rptt3 of the unsplit algorithm are not built: dim_split only)
rp_vc_acoustics_3d = RiemannSolver("vc_acoustics_3d", 20, 3, 4, 2, [])

_ALL = [rp_elasticity_fwave_1d, rp_psystem_fwave_2d, rp_advection_1d, rp_acoustics_1d, rp_advection_color_1d, rp_burgers_1d, rp_euler_1d, rp_shallow_1d, rp_advection_2d, rp_shallow_2d, rp_vc_acoustics_2d, rp_vc_advection_2d, rp_acoustics_2d, rp_euler_5wave_2d, rp_shallow_sphere_2d, rp_vc_acoustics_3d]
BY_NAME = dict((r.name, r) for r in _ALL)


def get(rp):
    if isinstance(rp, RiemannSolver):
        return rp
    if isinstance(rp, str):
        key = rp[3:] if rp.startswith("rp_") else rp
        if key in BY_NAME:
            return BY_NAME[key]
    raise Exception("Unknown Riemann solver %r; available: %s" % (rp, sorted(BY_NAME)))
