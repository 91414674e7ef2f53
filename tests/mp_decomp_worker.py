"""
Worker for tests/test_parallel_cpu.py (launched by torch.distributed.run, gloo, CPU only).
torch is TEST infrastructure here (the launcher and the host-array transport); the package itself imports none.

Runs a block-decomposed classic dim-split (or unsplit) computation and checks DECOMPOSITION
INVARIANCE: the gathered result must equal the serial run bit for bit (the reference asserts
serial == mpiexec -n 6 at 1e-14, test/test_examples.py:264-277).

What is under test is the PRODUCT's host-side parallel logic:
  pyclaw_amd.parallel   proc_grid / block ranges / neighbour ranks (periodic wrap rules)
  pcl_halo_region       strip geometry used by the device pack/unpack kernels (C ABI, host code)
  the wire protocol     "for d in W,E,S,N,SW,SE,NW,NE: send towards d, receive from opposite(d)"
  BC placement          physical BCs only on blocks that touch the domain edge, after the halo
The transport here is gloo on host arrays (the product's is RCCL on device buffers) and the
per-block arithmetic is the CPU oracle: neither is what is being checked.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import driver as D          # noqa: E402
from oracle import oracle as O          # noqa: E402
from pyclaw_amd import _lib, parallel   # noqa: E402

OPP = [1, 0, 3, 2, 7, 6, 5, 4]


def region(d, send, I, J, g):
    o = np.zeros(4, dtype=np.int32)
    _lib.check(_lib.lib().pcl_halo_region(d, int(send), I, J, g, _lib.i(o)))
    return (slice(o[0], o[0] + o[2]), slice(o[1], o[1] + o[3]))


def halo_exchange(qbc, nbr, g):
    """2-D blocks: the (i, j) plane.  3-D blocks cut in y and z: the (j, k) plane, whole x-rows as elements
    (what csrc/halo.hpp does with elem = I)."""
    lead = (slice(None),) * (qbc.ndim - 2)
    I, J = qbc.shape[-2:]
    reqs, recvs = [], []
    for d in range(8):
        if nbr[d] >= 0:
            buf = torch.from_numpy(np.ascontiguousarray(qbc[lead + region(d, True, I, J, g)]))
            reqs.append(dist.isend(buf, dst=int(nbr[d])))
        o = OPP[d]
        if nbr[o] >= 0:
            idx = lead + region(o, False, I, J, g)
            rb = torch.empty(qbc[idx].shape, dtype=torch.float64)
            reqs.append(dist.irecv(rb, src=int(nbr[o])))
            recvs.append((idx, rb))
    for r in reqs:
        r.wait()
    for idx, rb in recvs:
        qbc[idx] = rb.numpy()


def local_bcs(qbc, g, bc_lower, bc_upper, user_lower, dec, nglob):
    """solver.py:354-381 on a block: only sides on the physical boundary; a periodic side is a
    local copy only when this block spans the whole dimension."""
    for idim in range(qbc.ndim - 1):
        lo = dec.ranges[idim][0] == 0
        hi = dec.ranges[idim][1] == nglob[idim]
        whole = lo and hi
        v = np.rollaxis(qbc, idim + 1, 1)
        if lo:
            bc = bc_lower[idim]
            if bc == D.CUSTOM:
                user_lower(idim, 0.0, qbc, g)
            elif bc == D.OUTFLOW:
                for i in range(g):
                    v[:, i, ...] = v[:, g, ...]
            elif bc == D.REFLECTING:
                for i in range(g):
                    v[:, i, ...] = v[:, 2 * g - 1 - i, ...]
                    v[idim + 1, i, ...] = -v[idim + 1, 2 * g - 1 - i, ...]
            elif bc == D.PERIODIC and whole:
                v[:, :g, ...] = v[:, -2 * g:-g, ...]
        if hi:
            bc = bc_upper[idim]
            if bc == D.OUTFLOW:
                for i in range(g):
                    v[:, -i - 1, ...] = v[:, -g - 1, ...]
            elif bc == D.REFLECTING:
                for i in range(g):
                    v[:, -i - 1, ...] = v[:, -2 * g + i, ...]
                    v[idim + 1, -i - 1, ...] = -v[idim + 1, -2 * g + i, ...]
            elif bc == D.PERIODIC and whole:
                v[:, -g:, ...] = v[:, g:2 * g, ...]


def field(meqn, I, J):
    """deterministic global field q[m,i,j] (what every decomposition must reproduce)"""
    m, i, j = np.meshgrid(np.arange(meqn), np.arange(I), np.arange(J), indexing="ij")
    return np.asfortranarray(1000.0 * m + i + 0.001 * j + np.sin(0.1 * i * j))


def acoustics3d_case(nsteps):
    """3-D dim-split acoustics, blocks cut in y and z (parallel.Decomposition axes [1, 2]): gathered result ==
    serial result, bit for bit -- with the limiters off.  step3ds.f sweeps only the slices 0..m+1 (ONE ghost
    layer, step3ds.f:110-111), unlike step2ds.f which sweeps every ghost row: with a limiter (reach 2) the
    reference's own dimension-split 3-D step therefore depends on the decomposition at block faces (the second
    ghost layer is not x-swept before the y sweep reads it).  The product keeps the reference's per-block
    semantics; what must be, and is, invariant is the unlimited second-order scheme (reach 1)."""
    rank, size = parallel.rank(), parallel.world_size()
    be = O.COracle()
    p = D.acoustics3d_problem('hom', mx=14, my=12, mz=10)
    rng = np.random.default_rng(9)
    p.q[...] = rng.standard_normal(p.q.shape)
    p.aux[0] = 1.0 + rng.random(p.q.shape[1:])
    p.aux[1] = 0.5 + rng.random(p.q.shape[1:])
    p.bc_lower = p.aux_bc_lower = [D.PERIODIC, D.REFLECTING, D.PERIODIC]
    p.bc_upper = p.aux_bc_upper = [D.PERIODIC, D.OUTFLOW, D.PERIODIC]
    p.limiters = 0
    D.setup(p)
    g = p.mbc
    nglob = list(p.q.shape[1:])
    dec = parallel.Decomposition(nglob, size, rank)
    assert dec.axes == [1, 2] and dec.ranges[0] == (0, nglob[0])
    periodic = [p.bc_lower[k] == D.PERIODIC for k in range(3)]
    nbr = dec.neighbors(periodic)
    sl = tuple(slice(a, b) for a, b in dec.ranges)
    loc = [b - a for a, b in dec.ranges]

    def padded(arr, bl, bu, is_aux):
        out = np.zeros((arr.shape[0],) + tuple(n + 2 * g for n in loc), order="F")
        out[(slice(None),) + (slice(g, -g),) * 3] = arr[(slice(None),) + sl]
        halo_exchange(out, nbr, g)
        if is_aux:
            saved = (p.bc_lower, p.bc_upper)
        local_bcs3(out, g, bl, bu, dec, nglob, is_aux)
        return out

    def local_bcs3(qbc, g, bl, bu, dec, nglob, is_aux):
        for idim in range(3):
            lo = dec.ranges[idim][0] == 0
            hi = dec.ranges[idim][1] == nglob[idim]
            whole = lo and hi
            v = np.rollaxis(qbc, idim + 1, 1)
            if lo:
                if bl[idim] == D.REFLECTING:
                    for i in range(g):
                        v[:, i, ...] = v[:, 2 * g - 1 - i, ...]
                        if not is_aux:
                            v[idim + 1, i, ...] = -v[idim + 1, 2 * g - 1 - i, ...]
                elif bl[idim] == D.PERIODIC and whole:
                    v[:, :g, ...] = v[:, -2 * g:-g, ...]
            if hi:
                if bu[idim] == D.OUTFLOW:
                    for i in range(g):
                        v[:, -i - 1, ...] = v[:, -g - 1, ...]
                elif bu[idim] == D.PERIODIC and whole:
                    v[:, -g:, ...] = v[:, g:2 * g, ...]

    auxbc = padded(p.aux, p.aux_bc_lower, p.aux_bc_upper, True)
    q = np.array(p.q[(slice(None),) + sl], order="F")
    qbc = np.zeros((4,) + tuple(n + 2 * g for n in loc), order="F")
    dt = 0.02
    inner = (slice(None),) + (slice(g, -g),) * 3
    for _ in range(nsteps):
        qbc[inner] = q
        halo_exchange(qbc, nbr, g)
        local_bcs3(qbc, g, p.bc_lower, p.bc_upper, dec, nglob, False)
        qold = qbc.copy("F")
        cfl = 0.0
        for idir, qo in ((1, qold), (2, qbc), (3, qbc)):
            _, c1 = be.step3ds(p.rp, max(loc), g, loc[0], loc[1], loc[2], qo, qbc, auxbc, p.d[0], p.d[1], p.d[2],
                               dt, p.method, p.mthlim, idir)
            cfl = max(cfl, c1)
        cfl = parallel.allreduce_max_host(cfl)
        q = np.array(qbc[inner], order="F")
        dt = dt * 0.9 / cfl
    parts = [None] * size
    dist.gather_object((dec.ranges, q, dt), parts if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        full = np.zeros_like(p.q)
        for (rg, blk, dtk) in parts:
            full[(slice(None),) + tuple(slice(a, b) for a, b in rg)] = blk
            assert dtk == dt
        sq = np.array(p.q, order="F")
        sqbc = np.zeros((4,) + tuple(n + 2 * g for n in nglob), order="F")
        sdt = 0.02
        for _ in range(nsteps):
            sqbc[inner] = sq
            D.fill_ghosts(sqbc, g, p.bc_lower, p.bc_upper)
            sold = sqbc.copy("F")
            scfl = 0.0
            for idir, qo in ((1, sold), (2, sqbc), (3, sqbc)):
                _, c1 = be.step3ds(p.rp, max(nglob), g, nglob[0], nglob[1], nglob[2], qo, sqbc, p.auxbc, p.d[0],
                                   p.d[1], p.d[2], sdt, p.method, p.mthlim, idir)
                scfl = max(scfl, c1)
            sq = np.array(sqbc[inner], order="F")
            sdt = sdt * 0.9 / scfl
        ok = bool(np.array_equal(full, sq)) and sdt == dt
        print("RESULT case=acoustics3d size=%d dims=%s equal=%s maxdiff=%g" % (size, dec.dims, ok,
                                                                               np.abs(full - sq).max()))
    parallel.barrier()
    parallel.shutdown()
    sys.exit(0 if ok else 1)


def checkpoint_case(outdir):
    """every rank writes its block of a 37 x 29 state as a 'block' checkpoint, then reads the frame back
    (same decomposition) through Solution(frame, format='block')"""
    import pyclaw_amd as pyclaw
    rank = parallel.rank()
    grid = pyclaw.Grid([pyclaw.Dimension('x', 0., 1., 37), pyclaw.Dimension('y', -1., 1., 29)])
    st = pyclaw.State(grid, 3, 2)
    full, fa = field(3, 37, 29), field(2, 37, 29) * 0.5
    (i0, i1), (j0, j1) = st.decomp.ranges
    assert st.q.shape == (3, i1 - i0, j1 - j0)
    st.q[...] = full[:, i0:i1, j0:j1]
    st.aux[...] = fa[:, i0:i1, j0:j1]
    st.t = 0.625
    st.aux_global['gamma'] = 1.4
    pyclaw.Solution(st).write(7, outdir, format='block', write_aux=True)
    back = pyclaw.Solution(7, path=outdir, format='block', read_aux=True)
    ok = (np.array_equal(back.state.q, st.q) and np.array_equal(back.state.aux, st.aux) and back.t == 0.625
          and back.state.aux_global['gamma'] == 1.4)
    print("RESULT case=checkpoint rank=%d equal=%s" % (rank, ok))
    parallel.barrier()
    parallel.shutdown()
    sys.exit(0 if ok else 1)


def main():
    case = sys.argv[1]
    nsteps = int(sys.argv[2])
    # the TEST's transport for host arrays: gloo (torch.distributed), set up here.  The product's control plane
    # (pyclaw_amd.parallel: rendezvous, barrier, max/sum of host scalars) is its own stdlib TCP group and runs
    # beside it -- both are exercised by every case.
    dist.init_process_group(backend="gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    parallel.init()
    rank, size = parallel.rank(), parallel.world_size()
    assert (rank, size) == (dist.get_rank(), dist.get_world_size())
    # broadcast_bytes is what carries the 128-byte ncclUniqueId in a GPU run
    token = parallel.broadcast_bytes(bytes(range(128)) if rank == 0 else None, src=0)
    assert token == bytes(range(128))
    assert parallel.allreduce_sum_host([1.0, float(rank)]) == [float(size), size * (size - 1) / 2.0]
    if case.startswith("checkpoint:"):
        checkpoint_case(case.split(":", 1)[1])
    if case == "acoustics3d":
        acoustics3d_case(nsteps)
    be = O.COracle()

    if case == "euler":
        p = D.shockbubble_problem(mx=48, my=36, with_src=False)
    elif case == "acoustics_periodic":
        p = D.acoustics2d_problem(mx=40, my=44, bcs=([D.PERIODIC, D.REFLECTING], [D.PERIODIC, D.OUTFLOW]))
    elif case == "acoustics_periodic_xy":
        p = D.acoustics2d_problem(mx=36, my=40, bcs=([D.PERIODIC, D.PERIODIC], [D.PERIODIC, D.PERIODIC]))
    elif case == "euler_unsplit":
        p = D.shockbubble_problem(mx=48, my=36, with_src=False, dim_split=False, order_trans=2)
    else:
        raise SystemExit("unknown case")
    D.setup(p)
    g = p.mbc
    nglob = list(p.q.shape[1:])
    dec = parallel.Decomposition(nglob, size, rank)
    (i0, i1), (j0, j1) = dec.ranges
    periodic = [p.bc_lower[k] == D.PERIODIC for k in range(2)]
    nbr = dec.neighbors(periodic)
    mx, my = i1 - i0, j1 - j0
    meqn = p.q.shape[0]
    q = np.array(p.q[:, i0:i1, j0:j1], order="F")
    qbc = np.zeros((meqn, mx + 2 * g, my + 2 * g), order="F")
    dt = p.dt_initial
    for _ in range(nsteps):
        qbc[:, g:-g, g:-g] = q
        halo_exchange(qbc, nbr, g)
        local_bcs(qbc, g, p.bc_lower, p.bc_upper, p.user_bc_lower, dec, nglob)
        qold = qbc.copy("F")
        maxm = max(mx, my)
        if p.dim_split:
            _, cx = be.step2ds(p.rp, p.rp_params, maxm, g, mx, my, qold, qbc, None, p.d[0], p.d[1], dt,
                               p.method, p.mthlim, 1)
            _, cy = be.step2ds(p.rp, p.rp_params, maxm, g, mx, my, qbc, qbc, None, p.d[0], p.d[1], dt,
                               p.method, p.mthlim, 2)
            cfl = max(cx, cy)
        else:
            _, cfl = be.step2(p.rp, p.rp_params, maxm, g, mx, my, qold, qbc, None, p.d[0], p.d[1], dt,
                              p.method, p.mthlim)
        cfl = parallel.allreduce_max_host(cfl)          # petclaw/cfl.py:29-31
        q = np.array(qbc[:, g:-g, g:-g], order="F")
        dt = min(p.dt_max, dt * p.cfl_desired / cfl)    # accept unconditionally; dt follows the global CFL

    # gather on rank 0 and compare with the serial computation
    parts = [None] * size
    dist.gather_object((i0, i1, j0, j1, q, dt), parts if rank == 0 else None, dst=0)
    ok = True
    if rank == 0:
        full = np.zeros_like(p.q)
        for (a0, a1, b0, b1, blk, dtk) in parts:
            full[:, a0:a1, b0:b1] = blk
            assert dtk == dt
        # serial reference: same loop on one block
        sq = np.array(p.q, order="F")
        sqbc = np.zeros((meqn, nglob[0] + 2 * g, nglob[1] + 2 * g), order="F")
        sdt = p.dt_initial
        for _ in range(nsteps):
            sqbc[:, g:-g, g:-g] = sq
            D.fill_ghosts(sqbc, g, p.bc_lower, p.bc_upper, p.user_bc_lower, p.user_bc_upper, 0.0)
            sold = sqbc.copy("F")
            maxm = max(nglob)
            if p.dim_split:
                _, cx = be.step2ds(p.rp, p.rp_params, maxm, g, nglob[0], nglob[1], sold, sqbc, None, p.d[0],
                                   p.d[1], sdt, p.method, p.mthlim, 1)
                _, cy = be.step2ds(p.rp, p.rp_params, maxm, g, nglob[0], nglob[1], sqbc, sqbc, None, p.d[0],
                                   p.d[1], sdt, p.method, p.mthlim, 2)
                scfl = max(cx, cy)
            else:
                _, scfl = be.step2(p.rp, p.rp_params, maxm, g, nglob[0], nglob[1], sold, sqbc, None, p.d[0],
                                   p.d[1], sdt, p.method, p.mthlim)
            sq = np.array(sqbc[:, g:-g, g:-g], order="F")
            sdt = min(p.dt_max, sdt * p.cfl_desired / scfl)
        ok = bool(np.array_equal(full, sq)) and sdt == dt
        print("RESULT case=%s size=%d dims=%s equal=%s maxdiff=%g" % (case, size, dec.dims, ok,
                                                                     np.abs(full - sq).max()))
    parallel.barrier()
    parallel.shutdown()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
