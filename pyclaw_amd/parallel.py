r"""
Process group + block decomposition: the PetClaw side layer re-cast for one GPU per process.

Replaces (reference): PETSc ``DA.create(... sizes=grid.n, stencil_width=mbc, comm=COMM_WORLD)``
and ``getRanges()`` (src/petclaw/state.py:191-232).  The data path (ghost-cell exchange, CFL
all-reduce) is RCCL called from libpyclaw_amd on the solver's HIP stream (csrc/halo.hpp);
this module only does the host-side control plane:

* rank / world size from the launcher's environment (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK``),
* the px x py processor grid (same rule as PETSc's DMDA: px ~ sqrt(size*nx/ny), then the
  nearest divisor) and the index ranges of each block,
* the 8 neighbour ranks of a block (BOX stencil), with periodic wrap where the physical
  boundary condition is periodic,
* a tiny key/value rendezvous to hand rank 0's ncclUniqueId to the other ranks -- done with
  ``torch.distributed`` (gloo, CPU) because the launcher is ``torch.distributed.run``; torch
  is imported only when WORLD_SIZE > 1 and never touches device memory here.
"""
import math
import os

# direction order shared with csrc/halo.hpp
W, E, S, N, SW, SE, NW, NE = range(8)
_OFFSETS = [(-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]

_state = {"rank": 0, "size": 1, "initialized": False, "dist": None}


def init(backend="gloo"):
    """Join the process group described by the environment (no-op for a single process)."""
    if _state["initialized"]:
        return
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if size > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            dist.init_process_group(backend=backend, rank=rank, world_size=size)
        _state["dist"] = dist
        rank, size = dist.get_rank(), dist.get_world_size()
    _state.update(rank=rank, size=size, initialized=True)


def shutdown():
    dist = _state["dist"]
    if dist is not None and dist.is_initialized():
        dist.destroy_process_group()
    _state.update(rank=0, size=1, initialized=False, dist=None)


def rank():
    return _state["rank"]


def world_size():
    return _state["size"]


def local_rank():
    return int(os.environ.get("LOCAL_RANK", str(rank())))


def device_ordinal():
    """HIP device of this rank: LOCAL_RANK, unless PCL_FORCE_DEVICE pins every rank to one ordinal
    (diagnostics on a single-GPU box)."""
    forced = os.environ.get("PCL_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank()


def barrier():
    if _state["dist"] is not None:
        _state["dist"].barrier()


def broadcast_bytes(data, src=0):
    """Hand a small bytes object from rank src to everyone (ncclUniqueId distribution)."""
    dist = _state["dist"]
    if dist is None:
        return data
    box = [data]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def allreduce_max_host(value):
    """Host-side max all-reduce (bench timing, CPU tests).  The solver's CFL uses RCCL."""
    dist = _state["dist"]
    if dist is None:
        return value
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t[0])


def allreduce_sum_host(values):
    """Host-side sum all-reduce of a short list (output functionals at output times)."""
    dist = _state["dist"]
    if dist is None:
        return list(values)
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) for v in t]


def proc_grid(n_global, size):
    """Processor grid for `size` blocks: PETSc DMDA's default rule (DMSetUp_DA_2D).

    2-D: m = int(0.5 + sqrt(M*size/N)), lowered to the nearest divisor of size; n = size/m.
    For 8192^2 on 8 ranks this gives 2 x 4 (the C4 configuration of BASELINE.json).
    """
    if len(n_global) == 1:
        return [size]
    if len(n_global) == 2:
        M, Ncells = n_global
        m = int(0.5 + math.sqrt(float(M) * float(size) / float(Ncells)))
        m = max(1, min(m, size))
        while m > 0 and size % m:
            m -= 1
        m = max(m, 1)
        n = size // m
        if M < m or Ncells < n:
            raise Exception("Too many processes for a %dx%d grid" % (M, Ncells))
        return [m, n]
    raise NotImplementedError("3-D decomposition")


def block_range(n, p, c):
    """Cells [start,end) of block c of p along a dimension of n cells (PETSc: remainder first)."""
    base, rem = divmod(n, p)
    start = c * base + min(c, rem)
    return start, start + base + (1 if c < rem else 0)


class Decomposition(object):
    """This rank's block of a px x py decomposition of a global grid.

    1-D / 2-D grids are cut along all their dimensions; a 3-D grid is cut along y and z only (`axes` =
    [1, 2]): x-rows stay whole, so the halo is again an 8-neighbour BOX stencil, now in the (y, z) plane
    with whole rows as elements (csrc/halo.hpp)."""

    def __init__(self, n_global, size, rank, axes=None):
        self.n_global = list(n_global)
        self.size = size
        self.rank = rank
        ndim = len(self.n_global)
        self.axes = list(axes) if axes is not None else ([1, 2] if ndim == 3 else list(range(ndim)))
        self.dims = proc_grid([self.n_global[a] for a in self.axes], size)
        nd = len(self.dims)
        # rank = cx + px*cy  (first decomposed axis fastest, like DMDA)
        self.coords = [rank % self.dims[0]] if nd == 1 else [rank % self.dims[0], rank // self.dims[0]]
        self.ranges = [(0, n) for n in self.n_global]
        for k, a in enumerate(self.axes):
            self.ranges[a] = block_range(self.n_global[a], self.dims[k], self.coords[k])

    def rank_of(self, coords):
        return coords[0] if len(self.dims) == 1 else coords[0] + self.dims[0] * coords[1]

    def neighbors(self, periodic):
        """Ranks of the W,E,S,N,SW,SE,NW,NE neighbour blocks, -1 where there is none.

        A dimension held by a single block never has halo neighbours (its physical BC,
        periodic included, is a local ghost fill: solver.py:362-365); otherwise blocks at a
        physical edge wrap around only if that dimension's BC is periodic.
        """
        nd = len(self.dims)
        periodic = [periodic[a] for a in self.axes] if len(periodic) != nd else list(periodic)
        out = []
        for (ox, oy) in _OFFSETS:
            off = [ox] if nd == 1 else [ox, oy]
            if nd == 1 and oy != 0:
                out.append(-1)
                continue
            c = list(self.coords)
            ok = True
            for k in range(nd):
                if off[k] == 0:
                    continue
                if self.dims[k] == 1:
                    ok = False
                    break
                c[k] += off[k]
                if c[k] < 0 or c[k] >= self.dims[k]:
                    if periodic[k]:
                        c[k] %= self.dims[k]
                    else:
                        ok = False
                        break
            out.append(self.rank_of(c) if ok else -1)
        return out


def decompose(grid):
    """Called by State(): split `grid` over the process group (None if single process)."""
    init()
    if world_size() == 1:
        return None
    existing = getattr(grid, "_decomp", None)
    if existing is not None:
        return existing
    dec = Decomposition(grid.n, world_size(), rank())
    for k, dim in enumerate(grid.dimensions):
        dim._set_range(*dec.ranges[k])
    grid._decomp = dec
    return dec
