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


def test_six_ranks_equal_serial():
    """the reference's own parallel tests use mpiexec -n 6 (test/util.py:64-95)"""
    run_workers(6, "acoustics_periodic")
