r"""
State: q, aux and problem scalars on one grid (reference: src/pyclaw/state.py:10-236;
parallel twin src/petclaw/state.py:136-293).

``q`` and ``aux`` are host numpy arrays, Fortran ordered ``(meqn, nx[, ny])`` exactly as in
the reference; the HIP solver keeps its own resident copy in HBM while it is evolving and
writes the result back here at the end of ``evolve_to_time``.

When the process group has more than one rank (pyclaw_amd.parallel), constructing a State
decomposes its grid: every Dimension gets this rank's ``nstart/nend`` and ``q``/``aux`` hold
only the local block -- what PETSc's DMDA ranges did in petclaw (state.py:191-197).
"""
import numpy as np

from . import parallel
from .grid import Grid


class State(object):
    def __init__(self, grid, meqn, maux=0):
        if not isinstance(grid, Grid):
            raise Exception("""A PyClaw State object must be initialized with
                             a PyClaw Grid object.""")
        self.grid = grid
        self.p = None
        self.F = None
        self.aux_global = {}
        self.t = 0.
        self.mcapa = -1
        self.decomp = parallel.decompose(grid)   # None when single-process
        self.q = self.new_array(meqn)
        self.aux = self.new_array(maux)

    @property
    def meqn(self):
        if self.q is None:
            raise Exception('state.meqn has not been set.')
        return self.q.shape[0]

    @property
    def mp(self):
        return self.p.shape[0] if self.p is not None else 0

    @mp.setter
    def mp(self, mp):
        if self.p is not None:
            raise Exception('Cannot change state.mp after aux is initialized.')
        self.p = self.new_array(mp)

    @property
    def mF(self):
        return self.F.shape[0] if self.F is not None else 0

    @mF.setter
    def mF(self, mF):
        if self.F is not None:
            raise Exception('Cannot change state.mF after aux is initialized.')
        self.F = self.new_array(mF)

    @property
    def maux(self):
        return self.aux.shape[0] if self.aux is not None else 0

    def is_valid(self):
        return self.q is not None and self.q.flags['F_CONTIGUOUS']

    def set_mbc(self, mbc):
        """Virtual in pyclaw (state.py:164-168); the DMDA rebuild of petclaw has no analogue."""
        pass

    def set_q_from_qbc(self, mbc, qbc):
        """state.py:171-186"""
        ndim = self.grid.ndim
        if ndim == 1:
            self.q = qbc[:, mbc:-mbc]
        elif ndim == 2:
            self.q = qbc[:, mbc:-mbc, mbc:-mbc]
        elif ndim == 3:
            self.q = qbc[:, mbc:-mbc, mbc:-mbc, mbc:-mbc]
        else:
            raise Exception("Assumption (1 <= ndim <= 3) violated.")

    def get_qbc_from_q(self, mbc, whichvec, qbc):
        """state.py:188-206"""
        ndim = self.grid.ndim
        q = self.q if whichvec == 'q' else self.aux
        if ndim == 1:
            qbc[:, mbc:-mbc] = q
        elif ndim == 2:
            qbc[:, mbc:-mbc, mbc:-mbc] = q
        elif ndim == 3:
            qbc[:, mbc:-mbc, mbc:-mbc, mbc:-mbc] = q
        return qbc

    def new_array(self, dof):
        if dof == 0:
            return None
        shape = [dof]
        shape.extend(self.grid.ng)
        return np.empty(shape, order='F')

    def __deepcopy__(self, memo={}):
        import copy
        result = self.__class__.__new__(self.__class__)
        result.grid = self.grid            # grids are shared: the decomposition must stay identical
        result.p = None
        result.F = None
        result.t = copy.deepcopy(self.t)
        result.mcapa = self.mcapa
        result.decomp = self.decomp
        result.q = None if self.q is None else np.array(self.q, order='F', copy=True)
        result.aux = None if self.aux is None else np.array(self.aux, order='F', copy=True)
        result.aux_global = copy.deepcopy(self.aux_global)
        return result

    def __str__(self):
        output = "  t=%s meqn=%s\n  " % (self.t, self.meqn)
        if self.q is not None:
            output += "  q.shape=%s" % str(self.q.shape)
        if self.aux is not None:
            output += " aux.shape=%s" % str(self.aux.shape)
        return output
