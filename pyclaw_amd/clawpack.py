r"""
Classic Clawpack solvers on MI355X (reference: src/pyclaw/clawpack.py).

``ClawSolver1D`` / ``ClawSolver2D`` keep the reference's attributes and step structure
(start_step, Strang/Godunov source splitting, step_hyperbolic, CFL check: clawpack.py:114-165);
``step_hyperbolic`` calls the resident HIP path of libpyclaw_amd instead of the f2py modules
``classic1.step1`` / ``classic2.step2ds`` / ``classic2.step2`` (clawpack.py:323,538-552).

kernel_language: 'HIP' (default).  'Fortran' is accepted as an alias so reference scripts run
unchanged; 'Python' (the reference's pure-numpy 1-D kernel) is not provided.
"""
import numpy as np

from . import _lib, riemann
from .solver import Solver


class DeviceSource(object):
    """A source-term step (``solver.step_src``) that libpyclaw_amd applies on the device.

    A plain Python ``step_src(solver,state,dt)`` still works, but forces a device->host->device
    round trip of q per call (arbitrary numpy code cannot run on the GPU)."""

    def apply(self, solver, state, dt):
        raise NotImplementedError


class EulerRadialSource(DeviceSource):
    """Geometric source of the 2-D Euler equations with radial symmetry, 2-stage RK -- the
    ``euler_rad_src`` / ``step_Euler_radial`` callback of the reference's shock-bubble scripts
    (test/euler/2d/shockbubble.py:59-94).  aux[0] must hold the radial coordinate."""

    def __init__(self, gamma1, ndim=2):
        self.params = np.array([gamma1, float(ndim)], dtype=np.float64)

    def apply(self, solver, state, dt):
        _lib.check(_lib.lib().pcl_src(solver._h, 1, dt, _lib.d(self.params), 2))


class SphereCoriolisSource(DeviceSource):
    """Coriolis source of the shallow-water-on-the-sphere app (apps/shallow-sphere/src2.f, wrapped by
    ``fortran_src_wrapper``, shallow_4_Rossby_Haurwitz_wave.py:24-49): projection onto the tangent plane, 4-stage RK,
    projection.  aux[13:16] must hold the radial unit vector (setaux.f:150-156)."""

    def apply(self, solver, state, dt):
        _lib.check(_lib.lib().pcl_src(solver._h, 2, dt, None, 0))


class ClawSolver(Solver):
    r"""Generic classic Clawpack solver (clawpack.py:24-262)."""

    def __init__(self, data=None):
        for attr in ['limiters', 'order', 'src_split', 'fwave', 'step_src', 'start_step']:
            if attr not in self._required_attrs:
                self._required_attrs.append(attr)
        self._default_attr_values['mbc'] = 2
        self._default_attr_values['limiters'] = 1           # limiters.tvd.minmod
        self._default_attr_values['order'] = 2
        self._default_attr_values['src_split'] = 1
        self._default_attr_values['fwave'] = False
        self._default_attr_values['step_src'] = None
        self._default_attr_values['start_step'] = None
        self._default_attr_values['kernel_language'] = 'HIP'
        self._default_attr_values['verbosity'] = 0
        self._default_attr_values['cfl_max'] = 1.0
        self._default_attr_values['cfl_desired'] = 0.9
        # arithmetic mode of the HIP kernels: 'exact' (bit-identical to the reference's
        # no-FMA Fortran) or 'fast' (FMA + reciprocal-multiply divisions, rtol 1e-12)
        self._default_attr_values['math'] = 'exact'
        self.rp = None
        self._src_fused = False
        self._fuse_key = None
        self._rp_id = None
        self._cfl_out = None
        self._cfl_ptr = None
        super(ClawSolver, self).__init__(data)

    # ---------------------------------------------------------------- time stepping
    def step(self, solution):
        r"""clawpack.py:114-165"""
        state = solution.states[0]
        if self._fuse_key != (id(self.step_src), self.src_split):      # changed after setup(): decide again
            self._decide_src_fusion(state)
        if self.start_step is not None:
            self._pull(state)
            self.start_step(self, solution)
            self._push(state)

        if self.src_split == 2 and self.step_src is not None:
            self._apply_src(state, self.dt / 2.0)

        self.step_hyperbolic(solution)

        if self.cfl.get_cached_max() >= self.cfl_max:
            if self._src_fused and self.cfl.get_cached_max() == self.cfl_max:
                # The reference returns here WITHOUT the source term while evolve_to_time accepts cfl == cfl_max
                # (clawpack.py:153 vs solver.py:668): redo this step without the source fused into its last pass.
                L = _lib.lib()
                _lib.check(L.pcl_undo_step(self._h))
                _lib.check(L.pcl_fuse_source(self._h, 0, None, 0))
                try:
                    self.step_hyperbolic(solution)
                finally:
                    _lib.check(L.pcl_fuse_source(self._h, 1, _lib.d(self.step_src.params), 2))
            return False

        if self.step_src is not None:
            if self.src_split == 2:
                self._apply_src(state, self.dt / 2.0)
            if self.src_split == 1 and not self._src_fused:      # fused: the y pass applied it while storing
                self._apply_src(state, self.dt)
        return True

    def _apply_src(self, state, dt):
        if isinstance(self.step_src, DeviceSource):
            self.step_src.apply(self, state, dt)
            self._host_stale = True
        else:
            self._pull(state)
            self.step_src(self, state, dt)
            self._push(state)

    def _pre_step_modifies_q(self):
        return self.start_step is not None or (self.src_split == 2 and self.step_src is not None)

    def _backup(self, state):
        # Only code that changes q BEFORE the hyperbolic step needs a real copy; otherwise the
        # pre-step buffer survives the step and "restore" is a pointer swap.
        self._copied_backup = self._pre_step_modifies_q()
        if self._copied_backup:
            _lib.check(_lib.lib().pcl_backup(self._h))

    def _restore(self, state):
        if self._copied_backup:
            _lib.check(_lib.lib().pcl_restore(self._h))
        else:
            _lib.check(_lib.lib().pcl_undo_step(self._h))
        self._host_stale = True

    def check_cfl_settings(self):
        pass

    def step_hyperbolic(self, solution):
        r"""One homogeneous step on the resident state (clawpack.py:299-323,510-555)."""
        import ctypes
        state = solution.states[0]
        cfl = self._cfl_out
        if cfl is None:            # one c_double + its pointer for the solver's lifetime
            cfl = self._cfl_out = ctypes.c_double(0.0)
            self._cfl_ptr = ctypes.cast(ctypes.byref(cfl), _lib.dp)
        spec = self._device_bc_spec(state)
        if spec is not None:
            # every ghost fill runs on the device: BCs + step in one library call
            _lib.check(_lib.lib().pcl_bc_step(self._h, spec[2], spec[3], self.dt, self._cfl_ptr))
        else:
            self.apply_q_bcs(state)
            _lib.check(_lib.lib().pcl_step_hyperbolic(self._h, self.dt, self._cfl_ptr))
        self._host_stale = True
        self.cfl.update_global_max(cfl.value)

    def set_mthlim(self):
        r"""clawpack.py:181-190"""
        self.mthlim = self.limiters
        if not isinstance(self.limiters, list):
            self.mthlim = [self.mthlim]
        if len(self.mthlim) == 1:
            self.mthlim = self.mthlim * self.mwaves
        if len(self.mthlim) != self.mwaves:
            raise Exception('Length of solver.limiters is not equal to 1 or to solver.mwaves')

    def set_method(self, state):
        r"""clawpack.py:192-212"""
        self.method = np.empty(7, dtype=np.int32, order='F')
        self.method[0] = self.dt_variable
        self.method[1] = self.order
        if self.ndim == 1:
            self.method[2] = 0
        elif self.dim_split:
            self.method[2] = -1
        else:
            self.method[2] = self.order_trans
        self.method[3] = self.verbosity
        self.method[4] = 0
        self.method[5] = state.mcapa + 1
        self.method[6] = state.maux

    def _default_rp(self, state):
        raise Exception("solver.rp is not set: choose a Riemann solver from pyclaw_amd.riemann "
                        "(the reference links one per app Makefile)")

    def setup(self, solution):
        r"""clawpack.py:214-238: mbc, mthlim, method, cparam, BC arrays -- plus the device handle."""
        if self.kernel_language not in ('HIP', 'Fortran'):
            raise Exception("Unrecognized kernel_language; pyclaw_amd provides 'HIP' "
                            "('Fortran' is accepted as an alias)")
        state = solution.states[0]
        state.set_mbc(self.mbc)
        self.check_cfl_settings()
        self.set_mthlim()
        self.set_method(state)

        rp = riemann.get(self.rp) if self.rp is not None else self._default_rp(state)
        if rp.ndim != self.ndim:
            raise Exception("Riemann solver %s is %d-D but the solver is %d-D" % (rp.name, rp.ndim, self.ndim))
        if rp.mwaves != self.mwaves:
            raise Exception("solver.mwaves=%d but Riemann solver %s has %d waves" % (self.mwaves, rp.name, rp.mwaves))
        if rp.meqn != state.meqn:
            raise Exception("state.meqn=%d but Riemann solver %s has %d equations" % (state.meqn, rp.name, rp.meqn))
        params = rp.all_params(state)                 # the cparam common block (state.py:142-162)

        self._release()
        cfg = _lib.Config()
        cfg.ndim = self.ndim
        ng = state.grid.ng
        for k in range(self.ndim):
            cfg.n[k] = int(ng[k])
            cfg.d[k] = float(state.grid.d[k])
        cfg.mbc = self.mbc
        cfg.meqn = state.meqn
        cfg.mwaves = self.mwaves
        cfg.maux = state.maux
        for k in range(7):
            cfg.method[k] = int(self.method[k])
        for k, m in enumerate(self.mthlim):
            cfg.mthlim[k] = int(m)
        cfg.fwave = int(bool(self.fwave))
        cfg.rp = rp.id
        for k, v in enumerate(params):
            cfg.rp_params[k] = v
        from . import parallel
        cfg.device = parallel.device_ordinal() if state.decomp is not None else int(getattr(self, 'device', 0))
        if self.math not in ('exact', 'fast', 'strict'):
            raise Exception("solver.math must be 'exact', 'fast' or 'strict'")
        cfg.math = {'exact': 0, 'fast': 1, 'strict': 2}[self.math]
        import ctypes
        h = ctypes.c_void_p()
        _lib.check(_lib.lib().pcl_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h

        self.allocate_bc_arrays(state)
        self._setup_halo(state)
        self._upload_aux(state)
        self._rp_id = rp.id
        self._src_fused = False          # a fresh device handle
        self._decide_src_fusion(state)
        self._decide_exchange_ahead(state)

    def _decide_exchange_ahead(self, state):
        """Decomposed dimension-split 2-D runs: let the library send the new state's halo right behind the y pass
        (pcl_halo_exchange_ahead) when nothing between two steps touches q -- no start_step, no Strang half step, a
        source term only if it is fused into the y pass, every boundary condition a device one -- and EVERY rank's
        block qualifies (the ranks agree here: the order of operations on the communicator depends on the choice).
        Where every block can run the ONE-KERNEL step the default is the plain order instead -- exchange, then the
        step in its faster form (pcl_bc_step) -- which measured faster than that step's exchange-ahead on the quiet
        and on the dense state (4096^2 block, self-neighbours: 0.38 / 0.86 ms against 0.42 / 0.93);
        PCL_EXCHANGE_AHEAD=2 asks for its exchange-ahead order all the same, PCL_EXCHANGE_AHEAD=0 switches
        exchange-ahead off altogether."""
        import ctypes
        import os
        from . import parallel
        self.exchange_ahead = False
        if not self._halo_active:
            return
        yes = ctypes.c_int(0)
        _lib.check(_lib.lib().pcl_halo_can_overlap(self._h, ctypes.byref(yes)))
        knob = os.environ.get("PCL_EXCHANGE_AHEAD", "1")
        mine = (bool(yes.value) and knob != "0"
                and not self._pre_step_modifies_q() and (self.step_src is None or self._src_fused)
                and self._device_bc_spec(state) is not None and not state.grid.gauges)
        # 2: the block also has an interior box of one-kernel tiles.  The one-kernel step sends the new halo behind its
        # rim tiles, BEFORE the Courant number's all-reduce; the two-pass step after it: one order for the whole run
        codes = parallel.allgather(int(yes.value) if mine else 0)
        if all(c > 0 for c in codes):
            onek = all(c == 2 for c in codes)
            if onek and knob != "2":        # (the knob is part of the environment every rank was started with)
                return
            _lib.check(_lib.lib().pcl_halo_exchange_ahead(self._h, 2 if onek else 1))
            self.exchange_ahead = True

    def _decide_src_fusion(self, state):
        """Godunov-split device source of the 2-D Euler step: applied by the y pass / y phase while it stores its
        results (one read + write of q less per step; PCL_FUSE_SRC=0 keeps the separate source kernel).  Decided at
        setup and again by step() whenever step_src / src_split were changed afterwards."""
        import os
        fuse = (isinstance(self.step_src, EulerRadialSource) and self.src_split == 1 and self.ndim == 2
                and self._rp_id == 11 and state.mcapa < 0 and os.environ.get("PCL_FUSE_SRC", "1") != "0")
        if fuse:
            _lib.check(_lib.lib().pcl_fuse_source(self._h, 1, _lib.d(self.step_src.params), 2))
        elif self._src_fused:
            _lib.check(_lib.lib().pcl_fuse_source(self._h, 0, None, 0))
        self._src_fused = bool(fuse)
        self._fuse_key = (id(self.step_src), self.src_split)

    def teardown(self):
        super(ClawSolver, self).teardown()


class ClawSolver1D(ClawSolver):
    r"""Clawpack evolution routine in 1D (clawpack.py:268-406)."""

    def __init__(self, data=None):
        self.ndim = 1
        super(ClawSolver1D, self).__init__(data)


class ClawSolver2D(ClawSolver):
    r"""2D classic solver (clawpack.py:412-558): dimensional splitting or unsplit with
    transverse Riemann solves (``dim_split``, ``order_trans``)."""

    no_trans = 0
    trans_inc = 1
    trans_cor = 2

    def __init__(self, data=None):
        self._default_attr_values['dim_split'] = True
        self._default_attr_values['order_trans'] = self.trans_inc
        self.ndim = 2
        super(ClawSolver2D, self).__init__(data)

    def check_cfl_settings(self):
        if (not self.dim_split) and (self.order_trans == 0):
            cfl_recommended = 0.5
        else:
            cfl_recommended = 1.0
        if self.cfl_max > cfl_recommended:
            import warnings
            warnings.warn('cfl_max is set higher than the recommended value of %s' % cfl_recommended)
            warnings.warn(str(self.cfl_desired))


class ClawSolver3D(ClawSolver):
    r"""3D classic solver (clawpack.py:563-702): ``dim_split=True`` (Godunov splitting, ``step3ds``) or the unsplit
    algorithm with the ``rpt3``/``rptt3`` transverse solves (``step3``; ``order_trans`` 0, 10, 11, 20, 21, 22)."""

    no_trans = 0
    trans_inc = 11
    trans_cor = 22

    def __init__(self, data=None):
        self._default_attr_values['dim_split'] = True
        self._default_attr_values['order_trans'] = self.trans_cor
        self.ndim = 3
        super(ClawSolver3D, self).__init__(data)

    def setup(self, solution):
        super(ClawSolver3D, self).setup(solution)
