r"""
Clawpack ASCII frames: ``fort.tNNNN`` / ``fort.qNNNN`` / ``fort.aNNNN``
(reference: src/pyclaw/io/ascii.py:25-172 writer, :174-330 reader).

Byte-compatible with the reference writer (``%18.8e`` per value, one cell per line, blank line
after each row / plane) so the files can be read by existing Clawpack/VisClaw tooling.  This is
the SURVEY 8(f)3 row: it is not on the per-step path; ``Controller`` calls it at output times, after
the solver has written the device state back to ``state.q``.  One deliberate difference: the
reference hard-codes ``fort.q``/``fort.a`` whatever ``file_prefix`` is (ascii.py:73,78), so its
``compute_p`` output overwrites the q frame; here q/aux files honour ``file_prefix`` like the reader does.

In a decomposed run every rank writes its own block as ``fort.qNNNN.rank%04d`` with the block's
``lower`` / ``n`` in the header (the reference's parallel writer is PETSc binary, out of scope).
"""
import os

import numpy as np


def _rank_suffix(state):
    dec = getattr(state, 'decomp', None)
    return '' if dec is None else '.rank%04d' % dec.rank


def write_ascii(solution, frame, path, file_prefix='fort', write_aux=False, options={}, write_p=False):
    r"""Write fort.t, fort.q (and fort.a) for one frame (ascii.py:25-172)."""
    os.makedirs(path, exist_ok=True)
    state0 = solution.states[0]
    suffix = _rank_suffix(state0)
    meqn = state0.p.shape[0] if write_p else state0.meqn
    with open(os.path.join(path, '%s.t%s%s' % (file_prefix, str(frame).zfill(4), suffix)), 'w') as f:
        f.write("%18.8e     time\n" % solution.t)
        f.write("%5i                  meqn\n" % meqn)
        f.write("%5i                  nstates\n" % len(solution.states))
        f.write("%5i                  maux\n" % state0.maux)
        f.write("%5i                  ndim\n" % state0.grid.ndim)

    def header(fh, grid):
        fh.write("%5i                  grid_number\n" % grid.gridno)
        fh.write("%5i                  AMR_level\n" % grid.level)
        for dim in grid.dimensions:
            fh.write("%5i                  m%s\n" % (dim.ng, dim.name))
        for dim in grid.dimensions:
            fh.write("%18.8e     %slow\n" % (dim.lowerg, dim.name))
        for dim in grid.dimensions:
            fh.write("%18.8e     d%s\n" % (dim.d, dim.name))
        fh.write("\n")

    def body(fh, arr, ndim):
        nm = arr.shape[0]
        fmt = "%18.8e" * nm + "\n"
        if ndim == 1:
            for k in range(arr.shape[1]):
                fh.write(fmt % tuple(arr[:, k]))
        elif ndim == 2:
            for j in range(arr.shape[2]):
                for k in range(arr.shape[1]):
                    fh.write(fmt % tuple(arr[:, k, j]))
                fh.write('\n')
        elif ndim == 3:
            for l in range(arr.shape[3]):
                for j in range(arr.shape[2]):
                    for k in range(arr.shape[1]):
                        fh.write(fmt % tuple(arr[:, k, j, l]))
                    fh.write('\n')
                fh.write('\n')
        else:
            raise Exception("Dimension Exception in writing fort file.")

    with open(os.path.join(path, '%s.q%s%s' % (file_prefix, str(frame).zfill(4), suffix)), 'w') as q_file:
        for state in solution.states:
            header(q_file, state.grid)
            body(q_file, state.p if write_p else state.q, state.grid.ndim)
    if state0.maux > 0 and write_aux:
        with open(os.path.join(path, '%s.a%s%s' % (file_prefix, str(frame).zfill(4), suffix)), 'w') as aux_file:
            for state in solution.states:
                header(aux_file, state.grid)
                body(aux_file, state.aux, state.grid.ndim)


def _data_line(f, kind=float):
    return kind(f.readline().split()[0])


def read_ascii_t(frame, path='./', file_prefix='fort'):
    r"""[t, meqn, nstates, maux, ndim] of a frame (ascii.py:332-380)."""
    with open(os.path.join(path, '%s.t%s' % (file_prefix, str(frame).zfill(4)))) as f:
        t = _data_line(f)
        meqn = _data_line(f, int)
        nstates = _data_line(f, int)
        maux = _data_line(f, int)
        ndim = _data_line(f, int)
    return t, meqn, nstates, maux, ndim


def read_ascii(solution, frame, path='./', file_prefix='fort', read_aux=False, options={}):
    r"""Read a frame into `solution` (ascii.py:174-330): one State per grid in the file."""
    from ..grid import Dimension, Grid
    from ..state import State
    if frame < 0:
        raise IOError("Frame " + str(frame) + " does not exist ***")
    t, meqn, nstates, maux, ndim = read_ascii_t(frame, path, file_prefix)
    names = ['x', 'y', 'z']
    solution.states = []
    solution.grids = []

    def read_block(f, nm, with_data=True):
        gridno = _data_line(f, int)
        level = _data_line(f, int)
        n = [_data_line(f, int) for _ in range(ndim)]
        lower = [_data_line(f) for _ in range(ndim)]
        d = [_data_line(f) for _ in range(ndim)]
        f.readline()
        arr = np.empty([nm] + n, order='F')
        if ndim == 1:
            for i in range(n[0]):
                arr[:, i] = [float(v) for v in f.readline().split()]
        elif ndim == 2:
            for j in range(n[1]):
                for i in range(n[0]):
                    arr[:, i, j] = [float(v) for v in f.readline().split()]
                f.readline()
        else:
            for k in range(n[2]):
                for j in range(n[1]):
                    for i in range(n[0]):
                        arr[:, i, j, k] = [float(v) for v in f.readline().split()]
                    f.readline()
                f.readline()
        return gridno, level, n, lower, d, arr

    with open(os.path.join(path, '%s.q%s' % (file_prefix, str(frame).zfill(4)))) as f:
        for _ in range(nstates):
            gridno, level, n, lower, d, q = read_block(f, meqn)
            dims = [Dimension(names[i], lower[i], lower[i] + n[i] * d[i], n[i]) for i in range(ndim)]
            grid = Grid(dims)
            grid.gridno, grid.level = gridno, level
            state = State(grid, meqn, maux)
            state.t = t
            state.q[...] = q
            solution.states.append(state)
            solution.grids.append(grid)
    if read_aux and maux > 0:
        fname = os.path.join(path, '%s.a%s' % (file_prefix, str(frame).zfill(4)))
        if not os.path.exists(fname):
            fname = os.path.join(path, '%s.a0000' % file_prefix)    # time-independent aux (ascii.py:279-285)
        with open(fname) as f:
            for state in solution.states:
                state.aux[...] = read_block(f, maux)[5]
