r"""
Solver base class: the plugin surface of the reference (src/pyclaw/solver.py).

``evolve_to_time`` / ``step`` / ``setup`` / ``teardown`` keep the reference's semantics
(adaptive dt with CFL accept/reject and retake, solver.py:602-717; status dict; error
messages).  What differs is where the state lives: between the first and last line of
``evolve_to_time`` the authoritative ``q`` is the solver's resident copy in HBM; ``state.q`` on
the host is refreshed when control returns to the caller (and around user callbacks that
are plain Python).  The reference's ``q_backup = state.q.copy('F')`` (solver.py:660) becomes
a pointer swap on the device (``pcl_undo_step``) -- no copy at all in the common case.
"""
import logging

import numpy as np

from . import _lib, parallel
from .cfl import CFL


def default_compute_gauge_values(q, aux):
    r"""By default, record values of q at gauges (solver.py:26-29)."""
    return q


class BC():
    """Boundary condition types (solver.py:17-23)."""
    custom = 0
    outflow = 1
    periodic = 2
    reflecting = 3


class DeviceBC(object):
    """A user boundary condition that libpyclaw_amd applies on the device.

    Assign an instance to ``solver.user_bc_lower`` / ``user_bc_upper`` (with the matching
    ``bc_lower[idim] = BC.custom``) instead of a Python function to avoid moving ghost cells
    through the host every step.
    """

    def apply(self, solver, idim, side):
        raise NotImplementedError


class ConstantStateBC(DeviceBC):
    """Ghost cells of one side := a fixed state vector (e.g. the post-shock inflow state of
    test/euler/2d/shockbubble.py:41-57).  ``dims`` restricts it to some dimensions."""

    def __init__(self, state, dims=None):
        self.state = np.ascontiguousarray(state, dtype=np.float64)
        self.dims = dims

    def apply(self, solver, idim, side):
        if self.dims is not None and idim not in self.dims:
            return
        _lib.check(_lib.lib().pcl_bc_const(solver._h, idim, side, _lib.d(self.state)))


class SphereMirrorBC(DeviceBC):
    """The custom y boundary of the shallow-water-on-the-sphere app (``qbc_lower_y`` / ``qbc_upper_y``,
    apps/shallow-sphere/shallow_4_Rossby_Haurwitz_wave.py:295-313): ghost row j takes interior row 2*mbc-1-j with
    the x index reversed over the whole ghosted width, on the device."""

    def apply(self, solver, idim, side):
        if idim != 1:
            raise Exception("SphereMirrorBC is the y boundary of a 2-D grid")
        dec = getattr(solver, '_state', None) and solver._state.decomp
        if dec is not None and dec.dims[0] != 1:
            raise Exception("SphereMirrorBC reverses whole rows: decompose the grid in y only (PCL_PROC_GRID=1xN)")
        _lib.check(_lib.lib().pcl_bc(solver._h, idim, side, 4))


class Solver(object):
    r"""Pyclaw solver superclass; see the reference docstring (solver.py:25-125)."""

    _required_attrs = ['dt_initial', 'dt_max', 'cfl_max', 'cfl_desired', 'max_steps', 'dt_variable',
                       'mbc']
    _default_attr_values = {'dt_initial': 0.1, 'dt_max': 1e99, 'max_steps': 1000, 'dt_variable': True}

    def __init__(self, data=None):
        self.logger = logging.getLogger('evolve')
        # class-level defaults may be extended by subclasses before calling this
        for (k, v) in self._default_attr_values.items():
            self.__dict__.setdefault(k, v)
        if data is not None:
            for attr in self._required_attrs:
                if hasattr(data, attr):
                    setattr(self, attr, getattr(data, attr))

        self.dt = self._default_attr_values['dt_initial']
        self.cfl = CFL(self._default_attr_values['cfl_desired'])
        self.status = {'cflmax': self.cfl.get_cached_max(), 'dtmin': self.dt, 'dtmax': self.dt,
                       'numsteps': 0}
        self.bc_lower = [None] * self.ndim
        self.bc_upper = [None] * self.ndim
        self.aux_bc_lower = [None] * self.ndim
        self.aux_bc_upper = [None] * self.ndim
        self.user_bc_lower = None
        self.user_bc_upper = None
        self.user_aux_bc_lower = None
        self.user_aux_bc_upper = None
        self.compute_gauge_values = None
        self.qbc = None
        self.auxbc = None
        # device handle (libpyclaw_amd) and bookkeeping
        self._h = None
        self._resident = False       # True while the HBM copy is the authoritative q
        self._pinned = False         # begin_resident()/end_resident(): keep q in HBM across calls
        self._host_stale = False
        self._state = None

    # ------------------------------------------------------------------ validation
    def is_valid(self):
        valid = True
        for key in self._required_attrs:
            if key not in self.__dict__:
                self.logger.info('%s is not present.' % key)
                valid = False
        if any(b == BC.custom for b in self.bc_lower) and self.user_bc_lower is None:
            valid = False
        if any(b == BC.custom for b in self.bc_upper) and self.user_bc_upper is None:
            valid = False
        return valid

    def setup(self, solution):
        pass

    def teardown(self):
        # bookkeeping that dies with the handle: how many steps ran in each form of the dimension-split 2-D step
        # (one kernel / two passes; identical results, the library runs the faster one -- pcl_step_form_stats)
        if self._h is not None and hasattr(self, "status"):
            import ctypes
            ms, n, s1, s0 = ctypes.c_double(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
            if _lib.lib().pcl_step_form_stats(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(s1),
                                              ctypes.byref(s0)) == 0:
                self.status["step_forms"] = {"one_kernel": int(s1.value), "two_pass": int(s0.value)}
        self._release()

    def _release(self):
        if self._h is not None:
            _lib.lib().pcl_destroy(self._h)
            self._h = None
        self._resident = False

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def __str__(self):
        output = "Solver Status:\n"
        for (k, v) in self.status.items():
            output = "\n".join((output, "%s = %s" % (k.rjust(25), v)))
        return output

    # ------------------------------------------------------------------ host <-> device
    def _push(self, state):
        """state.q (host) -> HBM; the device copy becomes authoritative."""
        q = _lib.fortran64(state.q)
        _lib.check(_lib.lib().pcl_put_q(self._h, _lib.d(q), 0))
        self._resident = True
        self._host_stale = False
        self._state = state

    def _pull(self, state):
        """HBM -> state.q (host), if the host copy is stale."""
        if self._resident and self._host_stale:
            if not (state.q.flags.f_contiguous and state.q.dtype == np.float64 and state.q.flags.writeable):
                state.q = np.empty(state.q.shape, order='F')
            _lib.check(_lib.lib().pcl_get_q(self._h, _lib.d(state.q), 0))
            self._host_stale = False

    # ------------------------------------------------------------------ boundary conditions
    def allocate_bc_arrays(self, state):
        r"""qbc/auxbc host arrays with ghost cells (solver.py:297-313).  qbc is only a staging
        area for Python custom-BC callbacks here; auxbc is filled once and uploaded."""
        qbc_dim = [n + 2 * self.mbc for n in state.grid.ng]
        qbc_dim.insert(0, state.meqn)
        self.qbc = np.zeros(qbc_dim, order='F')
        if state.maux > 0:
            auxbc_dim = [n + 2 * self.mbc for n in state.grid.ng]
            auxbc_dim.insert(0, state.maux)
            self.auxbc = np.empty(auxbc_dim, order='F')
            self.apply_aux_bcs(state)
        else:
            self.auxbc = None

    def _at_lower(self, dim):
        return dim.nstart == 0

    def _at_upper(self, dim):
        return dim.nend == dim.n

    def apply_q_bcs(self, state):
        r"""Fill the ghost cells of the resident q (solver.py:315-381), same order: per
        dimension, lower then upper; only on blocks that touch the physical boundary.  In a
        decomposed run the halo exchange comes first (petclaw: globalToLocal inside
        get_qbc_from_q, src/petclaw/state.py:254-262)."""
        L = _lib.lib()
        if self._halo_active:
            _lib.check(L.pcl_halo_exchange(self._h))
        grid = state.grid
        for idim, dim in enumerate(grid.dimensions):
            whole = self._at_lower(dim) and self._at_upper(dim)
            if self._at_lower(dim):
                bc = self.bc_lower[idim]
                if bc == BC.custom:
                    self._custom_bc(state, dim, idim, 0, self.user_bc_lower)
                elif bc == BC.periodic:
                    if whole:
                        _lib.check(L.pcl_bc(self._h, idim, 0, bc))
                elif bc in (BC.outflow, BC.reflecting):
                    _lib.check(L.pcl_bc(self._h, idim, 0, bc))
                else:
                    raise NotImplementedError("Boundary condition %s not implemented" % bc)
            if self._at_upper(dim):
                bc = self.bc_upper[idim]
                if bc == BC.custom:
                    self._custom_bc(state, dim, idim, 1, self.user_bc_upper)
                elif bc == BC.periodic:
                    if whole:
                        _lib.check(L.pcl_bc(self._h, idim, 1, bc))
                elif bc in (BC.outflow, BC.reflecting):
                    _lib.check(L.pcl_bc(self._h, idim, 1, bc))
                else:
                    raise NotImplementedError("Boundary condition %s not implemented" % bc)

    def _device_bc_spec(self, state):
        """(types[2*ndim], constant states) for pcl_bc_step when every ghost fill of apply_q_bcs can
        run on the device (no plain-Python callback), else None.  Cached per solver setup."""
        key = (tuple(self.bc_lower), tuple(self.bc_upper), id(self.user_bc_lower), id(self.user_bc_upper))
        cached = getattr(self, '_bc_spec_cache', None)
        if cached is not None and cached[0] == key:
            return cached[1]
        types = np.full(2 * self.ndim, -1, dtype=np.int32)
        consts = np.zeros(2 * self.ndim * _lib.MAX_RP_PARAMS)
        spec = (types, consts)
        for idim, dim in enumerate(state.grid.dimensions):
            whole = self._at_lower(dim) and self._at_upper(dim)
            for side, at_edge, bcs, fn in ((0, self._at_lower(dim), self.bc_lower, self.user_bc_lower),
                                           (1, self._at_upper(dim), self.bc_upper, self.user_bc_upper)):
                if not at_edge:
                    continue
                bc = bcs[idim]
                k = 2 * idim + side
                if bc == BC.custom:
                    if isinstance(fn, ConstantStateBC) and state.meqn <= _lib.MAX_RP_PARAMS:
                        if fn.dims is None or idim in fn.dims:
                            types[k] = BC.custom
                            consts[k * _lib.MAX_RP_PARAMS:k * _lib.MAX_RP_PARAMS + state.meqn] = fn.state
                    elif (isinstance(fn, SphereMirrorBC) and self.ndim == 2 and idim == 1 and self._mirror_in_spec()
                          and (state.decomp is None or state.decomp.dims[0] == 1)):
                        types[k] = 4                       # PCL_BC_SPHERE_MIRROR
                    else:
                        spec = None
                elif bc == BC.periodic:
                    if whole:
                        types[k] = bc
                elif bc in (BC.outflow, BC.reflecting):
                    types[k] = bc
                else:
                    spec = None
        if spec is not None:
            # (types, consts, their ctypes pointers): the pointers are built once, not on every step
            spec = (types, consts, _lib.i(types), _lib.d(consts))
        self._bc_spec_cache = (key, spec)
        return spec

    def _mirror_in_spec(self):
        """the library takes the sphere app's pole boundary inside pcl_bc_step (unsplit classic step) and inside the
        SharpClaw stage calls; the dimension-split classic step evaluates its BCs in the x pass and does not"""
        return getattr(self, 'lim_type', None) is not None or not getattr(self, 'dim_split', True)

    def _custom_bc(self, state, dim, idim, side, fn):
        if fn is None:
            raise Exception("Custom BC requested but user_bc_%s is not set" % ("lower", "upper")[side])
        if isinstance(fn, DeviceBC):
            fn.apply(self, idim, side)
            return
        # Plain Python callback (state,dim,t,qbc,mbc): give it the reference's full qbc array
        # (solver.py:404-405), then send back only the ghost layers of this side.
        L = _lib.lib()
        _lib.check(L.pcl_get_q(self._h, _lib.d(self.qbc), 1))
        fn(state, dim, state.t, self.qbc, self.mbc)
        idx = [slice(None)] * self.qbc.ndim
        idx[idim + 1] = slice(0, self.mbc) if side == 0 else slice(self.qbc.shape[idim + 1] - self.mbc, None)
        strip = np.asfortranarray(self.qbc[tuple(idx)])
        _lib.check(L.pcl_put_strip(self._h, idim, side, self.mbc, _lib.d(strip)))

    def apply_aux_bcs(self, state):
        r"""Host-side aux ghost fill, once at setup (solver.py:456-596)."""
        self.auxbc = state.get_qbc_from_q(self.mbc, 'aux', self.auxbc)
        grid = state.grid
        mbc = self.mbc
        for idim, dim in enumerate(grid.dimensions):
            whole = self._at_lower(dim) and self._at_upper(dim)
            if self._at_lower(dim):
                bc = self.aux_bc_lower[idim]
                if bc == BC.custom:
                    self.user_aux_bc_lower(state, dim, state.t, self.auxbc, mbc)
                elif bc == BC.periodic and not whole:
                    pass
                else:
                    a = np.rollaxis(self.auxbc, idim + 1, 1)
                    if bc == BC.outflow:
                        for i in range(mbc):
                            a[:, i, ...] = a[:, mbc, ...]
                    elif bc == BC.periodic:
                        a[:, :mbc, ...] = a[:, -2 * mbc:-mbc, ...]
                    elif bc == BC.reflecting:
                        for i in range(mbc):
                            a[:, i, ...] = a[:, 2 * mbc - 1 - i, ...]
                    elif bc is None:
                        raise Exception("One or more of the aux boundary conditions aux_bc_upper has not been specified.")
                    else:
                        raise NotImplementedError("Boundary condition %s not implemented" % bc)
            if self._at_upper(dim):
                bc = self.aux_bc_upper[idim]
                if bc == BC.custom:
                    self.user_aux_bc_upper(state, dim, state.t, self.auxbc, mbc)
                elif bc == BC.periodic and not whole:
                    pass
                else:
                    a = np.rollaxis(self.auxbc, idim + 1, 1)
                    if bc == BC.outflow:
                        for i in range(mbc):
                            a[:, -i - 1, ...] = a[:, -mbc - 1, ...]
                    elif bc == BC.periodic:
                        a[:, -mbc:, ...] = a[:, mbc:2 * mbc, ...]
                    elif bc == BC.reflecting:
                        for i in range(mbc):
                            a[:, -i - 1, ...] = a[:, -2 * mbc + i, ...]
                    elif bc is None:
                        raise Exception("One or more of the aux boundary conditions aux_bc_lower has not been specified.")
                    else:
                        raise NotImplementedError("Boundary condition %s not implemented" % bc)

    def _upload_aux(self, state):
        """auxbc -> HBM.  Decomposed runs: exchange the aux halo, then redo the physical aux BCs on
        the device so that corner ghost cells next to a neighbour face are right (the reference
        order: globalToLocal, then auxbc_lower/upper; solver.py:492-523)."""
        if self.auxbc is None:
            return
        L = _lib.lib()
        _lib.check(L.pcl_put_aux(self._h, _lib.d(_lib.fortran64(self.auxbc))))
        if not self._halo_active:
            return
        _lib.check(L.pcl_halo_exchange_aux(self._h))
        for idim, dim in enumerate(state.grid.dimensions):
            whole = self._at_lower(dim) and self._at_upper(dim)
            for side, at_edge, bcs in ((0, self._at_lower(dim), self.aux_bc_lower),
                                       (1, self._at_upper(dim), self.aux_bc_upper)):
                if not at_edge:
                    continue
                bc = bcs[idim]
                if bc == BC.custom:
                    # the callback filled these ghost layers of the host auxbc (apply_aux_bcs), over the whole
                    # ghosted width of the block; they go onto the device at the callback's place in the reference
                    # order (exchange, then per dimension lower / upper).  Right as long as the callback computes
                    # them from positions alone, like the sphere app's setaux rows: one that copies interior values
                    # would miss the exchanged cells next to a neighbour face.
                    idx = [slice(None)] * self.auxbc.ndim
                    idx[idim + 1] = (slice(0, self.mbc) if side == 0
                                     else slice(self.auxbc.shape[idim + 1] - self.mbc, None))
                    strip = np.asfortranarray(self.auxbc[tuple(idx)])
                    _lib.check(L.pcl_put_aux_strip(self._h, idim, side, self.mbc, _lib.d(strip)))
                    continue
                if bc == BC.periodic and not whole:
                    continue
                _lib.check(L.pcl_bc_aux(self._h, idim, side, bc))

    # ------------------------------------------------------------------ multi-GPU glue
    _halo_active = False

    def _setup_halo(self, state):
        """Join the RCCL communicator if the state's grid is decomposed over several GPUs."""
        dec = state.decomp
        if dec is None:
            self._halo_active = False
            return
        L = _lib.lib()
        import ctypes as C
        import os
        # PetClaw's DMDA is periodic in every direction and the physical BCs overwrite the ghost cells of the sides
        # that are not (petclaw/state.py:205-208): a dimension wraps around as soon as EITHER of its sides is periodic
        # (the reference's 3-D heterogeneous test has reflecting lower and periodic upper sides)
        periodic = [self.bc_lower[k] == BC.periodic or self.bc_upper[k] == BC.periodic for k in range(state.grid.ndim)]
        nbr = np.array(dec.neighbors(periodic), dtype=np.int32)
        def host_wire():
            # host-staged wire: same device path, this module's TCP group instead of RCCL
            self._host_transport = parallel.host_transport()
            xfn, rfn = self._host_transport
            _lib.check(L.pcl_comm_init_host(self._h, parallel.world_size(), parallel.rank(), _lib.i(nbr),
                                            C.cast(xfn, C.c_void_p), C.cast(rfn, C.c_void_p), None))
            self._halo_active = True
            self.cfl._reduce = None

        if os.environ.get("PCL_HALO_TRANSPORT", "rccl") == "host":     # diagnostics; several ranks on ONE device
            self.halo_transport = "host"
            return host_wire()
        # RCCL.  Every rank learns whether EVERY rank got its communicator (a failure on one rank -- librccl missing,
        # ncclCommInitRank refusing -- must not leave the others waiting inside a collective).
        err = None
        uid = C.create_string_buffer(128)
        try:                                  # on every rank: loads librccl (rank 0's id is the one that is used)
            _lib.check(L.pcl_comm_unique_id(uid))
        except _lib.PclError as e:
            err = str(e)
        loaded = [e for e in parallel.allgather(err) if e]      # nobody enters ncclCommInitRank unless all can
        raw = parallel.broadcast_bytes((uid.raw if not loaded else b"") if parallel.rank() == 0 else None, src=0)
        if len(raw) != 128:
            err = err or loaded[0]
        else:
            try:
                _lib.check(L.pcl_comm_init(self._h, parallel.world_size(), parallel.rank(),
                                           C.create_string_buffer(raw, 128), _lib.i(nbr)))
            except _lib.PclError as e:
                err = str(e)
        errs = [e for e in parallel.allgather(err) if e]
        if errs:
            if os.environ.get("PCL_HALO_FALLBACK", "") != "host":
                raise Exception("RCCL communicator set-up failed on %d rank(s): %s  (PCL_HALO_FALLBACK=host would "
                                "run the same device path over the host-staged wire)" % (len(errs), errs[0]))
            self.logger.warning("RCCL set-up failed (%s): falling back to the host-staged halo wire" % errs[0])
            self.halo_transport = "host (fallback: RCCL set-up failed: %s)" % errs[0]
            return host_wire()
        self.halo_transport = "rccl"
        self._halo_active = True
        # the CFL all-reduce (petclaw/cfl.py:29-31) happens inside pcl_step_hyperbolic /
        # pcl_sharp_dq on the device: the value they return is already the global maximum
        self.cfl._reduce = None

    # ------------------------------------------------------------------ evolution
    def evolve_to_time(self, solution, tend=None):
        r"""Evolve solution from solution.t to tend (one step if tend is None); returns the
        status dict.  Line-for-line semantics of solver.py:602-717."""
        if self._h is None:
            raise Exception("solver.setup(solution) must be called before evolve_to_time")
        take_one_step = tend is None
        retake_step = False
        tstart = solution.t

        self.status['cflmax'] = self.cfl.get_cached_max()
        self.status['dtmin'] = self.dt
        self.status['dtmax'] = self.dt
        self.status['numsteps'] = 0

        if not self.dt_variable:
            if take_one_step:
                self.max_steps = 1
            else:
                self.max_steps = int((tend - tstart + 1e-10) / self.dt)
                if abs(self.max_steps * self.dt - (tend - tstart)) > 1e-5 * (tend - tstart):
                    raise Exception('dt does not divide (tend-tstart) and dt is fixed!')
        if self.dt_variable == 1 and self.cfl_desired > self.cfl_max:
            raise Exception('Variable time-stepping and desired CFL > maximum CFL')
        if (not take_one_step) and tend <= tstart:
            self.logger.info("Already at or beyond end time: no evolution required.")
            self.max_steps = 0

        state = solution.state
        if not self._pinned:
            self._push(state)
        try:
            for n in range(self.max_steps):
                state = solution.state
                if (not take_one_step) and solution.t + self.dt > tend and tstart < tend:
                    self.dt = tend - solution.t

                if self.dt_variable:
                    self._backup(state)          # q_backup = state.q.copy('F')
                    told = solution.t
                retake_step = False

                self.step(solution)

                cfl = self.cfl.get_cached_max()
                if cfl <= self.cfl_max:
                    self.status['cflmax'] = max(cfl, self.status['cflmax'])
                    if self.dt_variable == True:
                        solution.t += self.dt
                    else:
                        solution.t = tstart + (n + 1) * self.dt
                    if self.logger.isEnabledFor(logging.DEBUG):
                        self.logger.debug("Step %i  CFL = %f   dt = %f   t = %f" % (n, cfl, self.dt, solution.t))
                    self.write_gauge_values(solution)
                    self.status['numsteps'] += 1
                    if take_one_step or solution.t >= tend:
                        break
                else:
                    self.logger.debug("Rejecting time step, CFL number too large")
                    if self.dt_variable:
                        self._restore(state)     # state.q = q_backup
                        solution.t = told
                        retake_step = True
                    else:
                        self.status['cflmax'] = max(cfl, self.status['cflmax'])
                        raise Exception('CFL too large, giving up!')

                if self.dt_variable:
                    if cfl > 0.0:
                        self.dt = min(self.dt_max, self.dt * self.cfl_desired / cfl)
                        self.status['dtmin'] = min(self.dt, self.status['dtmin'])
                        self.status['dtmax'] = max(self.dt, self.status['dtmax'])
                    else:
                        self.dt = self.dt_max
        finally:
            if not self._pinned:
                self._pull(solution.state)
                self._resident = False

        if self.dt_variable and (not take_one_step) and solution.t < tend \
                and self.status['numsteps'] == self.max_steps:
            raise Exception("Maximum number of timesteps have been taken")
        return self.status

    def step(self, solution):
        raise NotImplementedError("No stepping routine has been defined!")

    # Extension of the reference surface: keep q resident in HBM across several
    # evolve_to_time calls (the reference's state.q is host memory, so every call would
    # otherwise upload at entry and download at exit).  state.q on the host is stale in
    # between; end_resident() refreshes it.
    def begin_resident(self, solution):
        self._push(solution.state)
        self._pinned = True

    def end_resident(self, solution):
        self._pull(solution.state)
        self._pinned = False
        self._resident = False

    # backup/restore of the resident state; subclasses pick the cheap form when legal
    def _backup(self, state):
        _lib.check(_lib.lib().pcl_backup(self._h))

    def _restore(self, state):
        _lib.check(_lib.lib().pcl_restore(self._h))
        self._host_stale = True

    # ------------------------------------------------------------------ gauges
    def write_gauge_values(self, solution):
        r"""solver.py:731-741.  While the state is resident the gauge cells are gathered on the device
        (pcl_get_cells: ngauges*(meqn+maux) doubles over PCIe) instead of pulling the whole q."""
        grid = solution.state.grid
        gauges = grid.gauges
        if not gauges:
            return
        state = solution.state
        compute = self.compute_gauge_values or default_compute_gauge_values
        if self._resident and self._h is not None:
            n = len(gauges)
            ij = np.zeros((n, 3 if state.grid.ndim > 2 else 2), dtype=np.int32)      # (i, j) pairs; (i, j, k) in 3-D
            for c, g in enumerate(gauges):
                ij[c, :len(g)] = g
            qv = np.empty((n, state.meqn))
            av = np.empty((n, max(state.maux, 1)))
            _lib.check(_lib.lib().pcl_get_cells(self._h, n, _lib.i(ij), _lib.d(qv),
                                                _lib.d(av) if state.maux > 0 else None))
            cells = [(qv[c], av[c, :state.maux]) for c in range(n)]
        else:
            cells = [(state.q[(slice(None),) + tuple(g)],
                      state.aux[(slice(None),) + tuple(g)] if state.aux is not None else None) for g in gauges]
        for i, (q, aux) in enumerate(cells):
            p = compute(q, aux)
            grid.gauge_files[i].write(str(solution.t) + ' ' + ' '.join(str(j) for j in p) + '\n')
