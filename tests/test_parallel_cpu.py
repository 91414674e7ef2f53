"""
CPU, multi-process (gloo): the N > 1 path's host logic -- decomposition, neighbour ranks, halo
strip geometry, send/recv protocol order, BCs only on boundary blocks -- must give results
identical to the serial run (decomposition invariance).
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_workers(nproc, case, nsteps=4):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "mp_decomp_worker.py"), case, str(nsteps)]
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert "equal=True" in out, out[-3000:]


@pytest.mark.parametrize("case", ["euler", "acoustics_periodic", "acoustics_periodic_xy", "euler_unsplit"])
def test_two_ranks_equal_serial(case):
    run_workers(2, case)


@pytest.mark.parametrize("case", ["euler", "acoustics_periodic_xy"])
def test_four_ranks_equal_serial(case):
    run_workers(4, case)


@pytest.mark.parametrize("nproc", [2, 4])
def test_3d_blocks_cut_in_y_and_z_equal_serial(nproc):
    """3-D dim-split acoustics: x-rows stay whole, blocks are cut in y and z (Decomposition axes [1, 2])"""
    run_workers(nproc, "acoustics3d", nsteps=3)


def test_six_ranks_equal_serial():
    """the reference's own parallel tests use mpiexec -n 6 (test/util.py:64-95)"""
    run_workers(6, "acoustics_periodic")


@pytest.mark.parametrize("nproc", [2, 4])
def test_checkpoint_written_by_n_ranks_restarts_on_one(nproc, tmp_path):
    """block checkpoints (pyclaw_amd/io/block.py).  N ranks write their blocks and read
    them back; then THIS single process reads the same frame and must get the whole field."""
    import json
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    run_workers(nproc, "checkpoint:" + str(tmp_path))
    import pyclaw_amd as pyclaw
    from mp_decomp_worker import field
    sol = pyclaw.Solution(7, path=str(tmp_path), format='block', read_aux=True)
    assert sol.state.decomp is None and sol.t == 0.625
    assert np.array_equal(sol.state.q, field(3, 37, 29))
    assert np.array_equal(sol.state.aux, field(2, 37, 29) * 0.5)
    hdr = json.load(open(os.path.join(str(tmp_path), "claw.ckpt0007.json")))
    assert len(hdr["blocks"]) == nproc and hdr["n"] == [37, 29]
    sizes = [os.path.getsize(os.path.join(str(tmp_path), b["file"])) for b in hdr["blocks"]]
    assert sum(sizes) == 8 * (3 + 2) * 37 * 29          # raw float64, nothing else in the block files
