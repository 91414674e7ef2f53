"""
GPU, several processes on ONE device: the decomposed device path end to end (tests/mp_gpu_worker.py) against the
serial oracle replay, bit for bit -- what a multi-GPU run does, except that the halo strips and the CFL maximum
travel through the host (PCL_HALO_TRANSPORT=host) instead of RCCL, which cannot put two ranks on one GPU.
At most 4 ranks (the box allows 6 processes on the card).
"""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(case, nranks, overlap=None):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "PCL_HALO_TRANSPORT": "host", "PCL_FORCE_DEVICE": "0",
                    "TORCHELASTIC_RUN_ID": "mp%d" % port})
        if overlap is not None:
            env["PCL_HALO_OVERLAP"] = str(overlap)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_gpu_worker.py"), case], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=240)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    assert codes == [0] * nranks, "exit codes %s\n%s" % (codes, "\n----\n".join(o[-1500:] for o in outs))
    assert "bit-identical: True" in outs[0], outs[0][-1500:]


@pytest.mark.parametrize("case,nranks", [("shockbubble_ds", 2), ("shockbubble_ds", 4), ("shockbubble_unsplit", 4),
                                         ("acoustics_ds", 4), ("acoustics_unsplit", 4), ("acoustics_sharp", 4),
                                         ("acoustics_unsplit", 3), ("acoustics3d_ds", 4), ("acoustics3d_ds", 2),
                                         ("acoustics3d_unsplit", 4), ("acoustics3d_unsplit", 2),
                                         ("rotating_classic", 4), ("rotating_classic", 2), ("rotating_sharpclaw", 4),
                                         ("acoustics_sharp9", 4), ("shockbubble_pycb", 2), ("shockbubble_pycb", 4),
                                         ("sphere_classic", 2), ("sphere_classic", 4),
                                         ("sphere_sharpclaw", 2), ("sphere_sharpclaw", 4),
                                         ("acoustics_odd", 5), ("acoustics_odd", 3), ("acoustics_odd", 4),
                                         ("acoustics_ds_mbc3", 2), ("acoustics_ds_mbc3", 4),
                                         ("advection1d", 2), ("advection1d", 3), ("acoustics1d_sharp", 4)])
def test_decomposed_device_run_equals_serial(case, nranks):
    launch(case, nranks)


@pytest.mark.parametrize("case", ["acoustics_ds", "acoustics_unsplit", "acoustics_sharp", "sphere_sharpclaw"])
def test_decomposed_device_run_sequential_exchange(case):
    """PCL_HALO_OVERLAP=0: the exchange in front of the step on one stream"""
    launch(case, 4, overlap=0)


@pytest.mark.parametrize("case,nranks", [("shockbubble_ds", 4), ("acoustics_ds", 4), ("acoustics_ds_mbc3", 4),
                                         ("shockbubble_unsplit", 4), ("acoustics_unsplit", 4), ("sphere_classic", 4),
                                         ("rotating_classic", 4)])
def test_decomposed_device_run_explicit_overlap(case, nranks):
    """PCL_HALO_OVERLAP=1 set explicitly: the two-pass dimension-split step with its interior x tiles beside the exchange
    (and exchange-ahead where the blocks agree on it) -- by default blocks of the aux-free solvers that cannot run the
    one-kernel step take the exchange in front of the step (pclaw.hip: twopass_overlap_ok) -- and the unsplit step with
    its interior x-phase tiles beside the exchange, which also runs only on request now (pcl_bc_step)"""
    launch(case, nranks, overlap=1)
