r"""
Process group + block decomposition: the PetClaw side layer re-cast for one GPU per process.

Replaces (reference): PETSc ``DA.create(... sizes=grid.n, stencil_width=mbc, comm=COMM_WORLD)``
and ``getRanges()`` (src/petclaw/state.py:191-232).  The data path (ghost-cell exchange, CFL
all-reduce) is RCCL called from libpyclaw_amd on the solver's HIP stream (csrc/halo.hpp);
this module only does the host-side control plane:

* rank / world size from the launcher's environment (``RANK``, ``WORLD_SIZE``, ``LOCAL_RANK`` --
  what ``torch.distributed.run``, ``mpirun`` wrappers or a plain shell loop export),
* the px x py processor grid (same rule as PETSc's DMDA: px ~ sqrt(size*nx/ny), then the
  nearest divisor) and the index ranges of each block,
* the 8 neighbour ranks of a block (BOX stencil), with periodic wrap where the physical
  boundary condition is periodic,
* a tiny rendezvous that hands rank 0's 128-byte ncclUniqueId to the other ranks, plus barrier / max /
  sum of a few host doubles (bench timing, output functionals).  Standard library only (one TCP
  connection per rank to rank 0): **no PyTorch anywhere in the package**, so ``librccl`` resolves to
  /opt/rocm's copy, not to a wheel's bundled one.

Rendezvous: rank 0 listens on an ephemeral port and publishes ``host:port`` in a small file whose name is
derived from ``MASTER_ADDR``/``MASTER_PORT`` (+ ``TORCHELASTIC_RUN_ID``) under ``PCL_RDZV_DIR`` (default: a
per-user 0700 directory, ``$XDG_RUNTIME_DIR`` or ``/tmp/pyclaw_amd-<uid>``: ranks of one node, which is what the
benchmark contract launches; the file is created O_EXCL|O_NOFOLLOW and readers reject one owned by another uid).  The launcher's own port is
never bound here -- ``torch.distributed.run`` keeps its agent store on it.  For several nodes either point
``PCL_RDZV_DIR`` at a shared directory or set ``PCL_RDZV_ADDR=host:port`` (rank 0 binds exactly that).
"""
import atexit
import json
import math
import os
import socket
import struct
import time

# direction order shared with csrc/halo.hpp
W, E, S, N, SW, SE, NW, NE = range(8)
_OFFSETS = [(-1, 0), (1, 0), (0, -1), (0, 1), (-1, -1), (1, -1), (-1, 1), (1, 1)]

_state = {"rank": 0, "size": 1, "initialized": False, "group": None}

_MAGIC = "pyclaw_amd-rdzv-1"
_TIMEOUT = float(os.environ.get("PCL_RDZV_TIMEOUT", "300"))


def _rdzv_dir():
    """Directory of the rendezvous file: PCL_RDZV_DIR if set (a shared directory for several nodes), else a per-user
    directory nobody else can write to -- $XDG_RUNTIME_DIR, or /tmp/pyclaw_amd-<uid> created 0700 and checked to be ours
    (a world-writable /tmp would let another user of the node plant or symlink the file)."""
    d = os.environ.get("PCL_RDZV_DIR")
    if d:
        return d
    d = os.environ.get("XDG_RUNTIME_DIR")
    if d and os.path.isdir(d) and os.access(d, os.W_OK):
        return d
    d = os.path.join("/tmp", "pyclaw_amd-%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("pyclaw_amd.parallel: %s must be a directory owned by uid %d with mode 0700 "
                           "(or set PCL_RDZV_DIR)" % (d, os.getuid()))
    return d


def _send(sock, obj):
    raw = json.dumps(obj).encode("utf-8")
    sock.sendall(struct.pack("!I", len(raw)) + raw)


def _recv(sock):
    def exactly(n):
        buf = b""
        while len(buf) < n:
            chunk = sock.recv(n - len(buf))
            if not chunk:
                raise ConnectionError("pyclaw_amd.parallel: peer closed the rendezvous connection")
            buf += chunk
        return buf
    (n,) = struct.unpack("!I", exactly(4))
    return json.loads(exactly(n).decode("utf-8"))


class _Group(object):
    """Star-shaped process group: every rank keeps one TCP connection to rank 0.  The only primitive is
    allgather of a small JSON value (floats survive exactly: json writes repr()); barrier, broadcast and the
    reductions are built on it and reduce in rank order, so every rank computes the same bits."""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size
        self.peers = []          # rank 0: sockets of ranks 1..size-1 (index r-1)
        self.up = None           # other ranks: socket to rank 0
        self.addr_file = None
        explicit = os.environ.get("PCL_RDZV_ADDR")
        if explicit:
            host, port = explicit.rsplit(":", 1)
            port = int(port)
        else:
            host, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), 0
            tag = "%s_%s_%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ.get("MASTER_PORT", "0"),
                                os.environ.get("TORCHELASTIC_RUN_ID", "none"))
            tag = "".join(c if c.isalnum() or c in "._-" else "_" for c in tag)
            self.addr_file = os.path.join(_rdzv_dir(), "pyclaw_amd_rdzv_%s.addr" % tag)
        if rank == 0:
            self._serve(host, port, explicit is not None)
        else:
            self._join(host, port, explicit is not None)

    def _serve(self, host, port, explicit):
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        try:
            srv.bind((host, port))
        except OSError:
            srv.bind(("0.0.0.0", port))       # MASTER_ADDR may be a name that is not a local interface
        srv.listen(self.size)
        port = srv.getsockname()[1]
        if not explicit:
            tmp = self.addr_file + ".%d" % os.getpid()
            try:
                os.unlink(tmp)                   # a leftover of a dead process with the same pid
            except OSError:
                pass
            # never through a symlink, never onto an existing file
            fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
            with os.fdopen(fd, "w") as f:
                f.write("%s:%d:%d" % (host, port, os.getpid()))
            os.replace(tmp, self.addr_file)      # atomic: a reader never sees a partial file
            atexit.register(self._unlink)
        srv.settimeout(_TIMEOUT)
        got = {}
        while len(got) < self.size - 1:
            conn, _ = srv.accept()
            conn.settimeout(_TIMEOUT)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            hello = _recv(conn)
            ok = (isinstance(hello, dict) and hello.get("magic") == _MAGIC and hello.get("size") == self.size
                  and isinstance(hello.get("rank"), int) and 0 < hello["rank"] < self.size
                  and hello["rank"] not in got)
            _send(conn, {"ok": bool(ok)})
            if ok:
                got[hello["rank"]] = conn
            else:
                conn.close()               # a stray client (or a rank of another job): keep waiting
        srv.close()
        self.peers = [got[r] for r in range(1, self.size)]

    def _join(self, host, port, explicit):
        deadline = time.time() + _TIMEOUT
        last = None
        while time.time() < deadline:
            try:
                if not explicit:
                    fd = os.open(self.addr_file, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
                    with os.fdopen(fd) as f:
                        st = os.fstat(f.fileno())
                        if st.st_uid != os.getuid():       # somebody else's file: not this job's rank 0
                            raise ValueError("rendezvous file %s is owned by uid %d" % (self.addr_file, st.st_uid))
                        h, p, _pid = f.read().strip().split(":")
                    host, port = h, int(p)
                sock = socket.create_connection((host, port), timeout=5.0)
                sock.settimeout(_TIMEOUT)
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                _send(sock, {"magic": _MAGIC, "rank": self.rank, "size": self.size})
                if _recv(sock).get("ok"):
                    self.up = sock
                    return
                sock.close()
            except (OSError, ValueError, ConnectionError) as e:   # no file yet, stale file, server not up yet
                last = e
            time.sleep(0.05)
        raise RuntimeError("pyclaw_amd.parallel: rank %d could not reach rank 0 within %.0f s (%s)"
                           % (self.rank, _TIMEOUT, last))

    def _unlink(self):
        try:
            if self.addr_file:
                os.unlink(self.addr_file)
        except OSError:
            pass

    def allgather(self, value):
        if self.rank == 0:
            vals = [value] + [_recv(c) for c in self.peers]
            for c in self.peers:
                _send(c, vals)
            return vals
        _send(self.up, value)
        return _recv(self.up)

    def close(self):
        for c in self.peers + ([self.up] if self.up else []):
            try:
                c.close()
            except OSError:
                pass
        self._unlink()


def init(backend=None):
    """Join the process group described by the environment (no-op for a single process).  `backend` is accepted
    for source compatibility and ignored: the control plane is this module's own TCP rendezvous."""
    if _state["initialized"]:
        return
    size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if size < 1 or rank < 0 or rank >= size:
        raise RuntimeError("pyclaw_amd.parallel: bad RANK=%d / WORLD_SIZE=%d in the environment" % (rank, size))
    group = _Group(rank, size) if size > 1 else None
    _state.update(rank=rank, size=size, initialized=True, group=group)


def shutdown():
    g = _state["group"]
    if g is not None:
        try:
            g.allgather(None)      # nobody closes while a peer still talks
        except (OSError, ConnectionError):
            pass
        g.close()
    _state.update(rank=0, size=1, initialized=False, group=None)


def rank():
    return _state["rank"]


def world_size():
    return _state["size"]


def local_rank():
    return int(os.environ.get("LOCAL_RANK", str(rank())))


def device_ordinal():
    """HIP device of this rank: LOCAL_RANK, unless PCL_FORCE_DEVICE pins every rank to one ordinal
    (diagnostics on a single-GPU box; pcl_comm_init refuses it for more than one rank)."""
    forced = os.environ.get("PCL_FORCE_DEVICE")
    return int(forced) if forced is not None else local_rank()


def barrier():
    g = _state["group"]
    if g is not None:
        g.allgather(None)


def broadcast_bytes(data, src=0):
    """Hand a small bytes object from rank src to everyone (ncclUniqueId distribution)."""
    g = _state["group"]
    if g is None:
        return data
    vals = g.allgather(bytes(data).hex() if g.rank == src else None)
    return bytes.fromhex(vals[src])


def allgather(value):
    """Every rank's small JSON-able value, in rank order (a list of one for a single process)."""
    g = _state["group"]
    if g is None:
        return [value]
    return g.allgather(value)


def allreduce_max_host(value):
    """Host-side max all-reduce (bench timing, CPU tests).  The solver's CFL uses RCCL."""
    g = _state["group"]
    if g is None:
        return value
    return max(float(v) for v in g.allgather(float(value)))


_OPP = [1, 0, 3, 2, 7, 6, 5, 4]      # W,E,S,N,SW,SE,NW,NE -> opposite (csrc/halo.hpp)


def host_transport():
    """ctypes callbacks for pcl_comm_init_host (PCL_HALO_TRANSPORT=host): the packed halo strips and the CFL maximum
    travel over this module's TCP group instead of RCCL.  Diagnostics and multi-process tests on ONE device only
    (every exchange is an allgather of all strips through rank 0).  Returns (exchange_cb, reduce_cb): keep them alive."""
    import base64
    import ctypes as C
    lp = C.POINTER(C.c_long)
    ip = C.POINTER(C.c_int)
    dp = C.POINTER(C.c_double)
    XFN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, lp, lp, ip, C.c_int)
    RFN = C.CFUNCTYPE(C.c_int, C.c_void_p, dp)

    def exchange(user, send, recv, off, cnt, nbr, nm):
        try:
            g = _state["group"]
            me = rank()
            out = {}
            for d in range(8):
                if nbr[d] >= 0:
                    a, n = off[d] * nm, cnt[d] * nm
                    raw = C.string_at(C.addressof(send.contents) + 8 * a, 8 * n)
                    out["%d" % d] = base64.b64encode(raw).decode("ascii")
            allm = g.allgather({"from": me, "msgs": out})
            by_rank = dict((m["from"], m["msgs"]) for m in allm)
            for o in range(8):
                if nbr[o] < 0:
                    continue
                # the strip for my side o is what the block on that side sent towards opposite(o), i.e. towards me
                raw = base64.b64decode(by_rank[int(nbr[o])]["%d" % _OPP[o]])
                if len(raw) != 8 * cnt[o] * nm:
                    return 2
                C.memmove(C.addressof(recv.contents) + 8 * off[o] * nm, raw, len(raw))
            return 0
        except Exception:            # never let an exception cross the C boundary
            import traceback
            traceback.print_exc()
            return 1

    def reduce_max(user, value):
        try:
            value[0] = allreduce_max_host(value[0])
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1

    return XFN(exchange), RFN(reduce_max)


def allreduce_sum_host(values):
    """Host-side sum all-reduce of a short list (output functionals at output times); summed in rank order."""
    g = _state["group"]
    if g is None:
        return list(values)
    rows = g.allgather([float(v) for v in values])
    out = [0.0] * len(rows[0])
    for row in rows:
        for k, v in enumerate(row):
            out[k] += v
    return out


def proc_grid(n_global, size):
    """Processor grid for `size` blocks: PETSc DMDA's default rule (DMSetUp_DA_2D).

    2-D: m = int(0.5 + sqrt(M*size/N)), lowered to the nearest divisor of size; n = size/m.
    For 8192^2 on 8 ranks this gives 2 x 4 (the C4 configuration of BASELINE.json).
    """
    if len(n_global) == 1:
        return [size]
    if len(n_global) == 2:
        M, Ncells = n_global
        forced = os.environ.get("PCL_PROC_GRID")      # "PXxPY" (PETSc: -da_processors_x / _y)
        if forced:
            m, n = (int(v) for v in forced.lower().split("x"))
            if m * n != size or M < m or Ncells < n:
                raise Exception("PCL_PROC_GRID=%s does not fit %d processes on a %dx%d grid" % (forced, size, M, Ncells))
            return [m, n]
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
    if len(grid.n) == 1 and (os.environ.get("PCL_DECOMPOSE_1D", "1") == "0" or grid.n[0] < 8 * world_size()):
        # PetClaw cuts 1-D grids too (petclaw/state.py:199-234), and so does this layer since round 3 (W / E strips only).
        # Replicas -- every rank keeps the whole row -- remain for PCL_DECOMPOSE_1D=0 and for rows too short to give
        # every rank a few cells beyond its ghost width.
        return None
    dec = Decomposition(grid.n, world_size(), rank())
    for k, dim in enumerate(grid.dimensions):
        dim._set_range(*dec.ranges[k])
    grid._decomp = dec
    return dec
