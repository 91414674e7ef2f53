r"""
Checkpoint / restart frames in a parallel-friendly format.

Replaces (reference) ``petclaw/io/petsc.py:32-249`` -- a pickle header + PETSc Vec binary written
collectively -- with plain files every rank writes on its own:

    <prefix>.ckptNNNN.json          header, written by rank 0: t, meqn, maux, global grid, aux_global,
                                    and for every block its index ranges, file name and byte offsets
    <prefix>.ckptNNNN.bRRRR.bin     block of rank RRRR: raw little-endian float64, q then aux, each in
                                    Fortran order (component fastest, then x, y[, z]) = the layout of state.q

No pickle anywhere (nothing executable in a checkpoint).  Reading does not depend on the decomposition the
frame was written with: every rank memory-maps the block files that overlap its own index ranges and
copies the overlap, so a frame written by 8 ranks restarts on 1, 2 or 6.

Like the reference's frames, a checkpoint doubles as a restart file: ``Solution(frame, path=...,
format='block')`` + ``Controller.start_frame`` (controller.py:77,216).
"""
import json
import os

import numpy as np


def _names(prefix, frame):
    base = '%s.ckpt%s' % (prefix, str(frame).zfill(4))
    return base + '.json', base + '.b%04d.bin'


def write_block(solution, frame, path, file_prefix='claw', write_aux=False, options={}, write_p=False):
    from .. import parallel
    os.makedirs(path, exist_ok=True)
    state = solution.states[0]
    grid = state.grid
    dec = getattr(state, 'decomp', None)
    rank = dec.rank if dec is not None else 0
    hname, bname = _names(file_prefix or 'claw', frame)
    q = np.asfortranarray(state.p if write_p else state.q, dtype='<f8')
    with_aux = bool(write_aux) and state.maux > 0
    with open(os.path.join(path, bname % rank), 'wb') as f:
        f.write(q.tobytes(order='F'))
        if with_aux:
            f.write(np.asfortranarray(state.aux, dtype='<f8').tobytes(order='F'))
    if rank == 0:
        if dec is None:
            blocks = [{"rank": 0, "ranges": [[0, n] for n in grid.n]}]
        else:
            blocks = []
            for r in range(dec.size):
                d = parallel.Decomposition(dec.n_global, dec.size, r)
                blocks.append({"rank": r, "ranges": [list(rg) for rg in d.ranges]})
        for b in blocks:
            b["file"] = bname % b["rank"]
        header = {
            "format": "pyclaw_amd block checkpoint", "version": 1, "frame": int(frame), "t": float(solution.t),
            "meqn": int(q.shape[0]), "maux": int(state.maux), "with_aux": with_aux, "ndim": grid.ndim,
            "names": list(grid.name), "n": [int(v) for v in grid.n], "lower": [float(v) for v in grid.lower],
            "upper": [float(v) for v in grid.upper], "dtype": "<f8", "order": "F",
            "aux_global": {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in state.aux_global.items()
                           if isinstance(v, (int, float, str, bool, list, tuple, np.ndarray, np.floating, np.integer))},
            "mcapa": int(state.mcapa), "blocks": blocks,
        }
        tmp = os.path.join(path, hname + '.tmp')
        with open(tmp, 'w') as f:
            json.dump(header, f, indent=1, default=float)
        os.replace(tmp, os.path.join(path, hname))      # the header appears last and atomically
    parallel.barrier()


def read_block(solution, frame, path='./', file_prefix='claw', read_aux=True, options={}):
    from ..grid import Dimension, Grid
    from ..state import State
    hname, _ = _names(file_prefix or 'claw', frame)
    with open(os.path.join(path, hname)) as f:
        h = json.load(f)
    if h.get("format") != "pyclaw_amd block checkpoint":
        raise IOError("%s is not a pyclaw_amd block checkpoint" % hname)
    ndim, meqn, maux = h["ndim"], h["meqn"], h["maux"]
    dims = [Dimension(h["names"][k], h["lower"][k], h["upper"][k], h["n"][k]) for k in range(ndim)]
    grid = Grid(dims)
    state = State(grid, meqn, maux)          # decomposes over the CURRENT process group
    state.t = h["t"]
    state.mcapa = h.get("mcapa", -1)
    state.aux_global.update(h.get("aux_global", {}))
    mine = [(grid.nstart[k], grid.nend[k]) for k in range(ndim)]
    want_aux = read_aux and maux > 0 and h["with_aux"]
    for b in h["blocks"]:
        rg = b["ranges"]
        ov = [(max(mine[k][0], rg[k][0]), min(mine[k][1], rg[k][1])) for k in range(ndim)]
        if any(lo >= hi for lo, hi in ov):
            continue
        shape = [rg[k][1] - rg[k][0] for k in range(ndim)]
        ncell = int(np.prod(shape))
        fname = os.path.join(path, b["file"])
        src = tuple(slice(ov[k][0] - rg[k][0], ov[k][1] - rg[k][0]) for k in range(ndim))
        dst = tuple(slice(ov[k][0] - mine[k][0], ov[k][1] - mine[k][0]) for k in range(ndim))
        qm = np.memmap(fname, dtype='<f8', mode='r', offset=0, shape=tuple([meqn] + shape), order='F')
        state.q[(slice(None),) + dst] = qm[(slice(None),) + src]
        del qm
        if want_aux:
            am = np.memmap(fname, dtype='<f8', mode='r', offset=8 * meqn * ncell, shape=tuple([maux] + shape),
                           order='F')
            state.aux[(slice(None),) + dst] = am[(slice(None),) + src]
            del am
    solution.states = [state]
    solution.grids = [grid]
