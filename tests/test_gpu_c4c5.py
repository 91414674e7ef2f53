"""
GPU: BASELINE configs[3] (C4, Euler 8192^2 on 2 x 4 blocks) and configs[4] (C5, SharpClaw WENO5 on the sphere,
2048 x 1024 on 4 blocks) at their sizes through the decomposed device path.

Per case (tests/mp_fullsize_worker.py describes them):
  1. the whole grid on one GPU through Controller.run (the serial run);
  2. ORACLE WINDOWS of that result: windows that straddle the faces and corner junctions of the block layout (and the
     periodic wrap) are replayed by the CPU oracle from the same initial cells -- all steps, with the context the
     stencil needs -- and must equal the GPU result bit for bit;
  3. four processes sharing the GPU (host-staged halo wire: RCCL refuses two ranks per device; the box allows 6
     processes on the card, so 8 ranks of 4096 x 2048 cannot run here -- the 2 x 2 layouts below exercise the same
     block shape with all eight neighbours) run the same case decomposed; every block's hash must equal the hash of
     the same cells of the serial result, and step count / Courant number / final dt must be identical:
     decomposed == serial, bit for bit (the reference's acceptance test for petclaw, test/test_examples.py:264-277).
Reference counterparts: src/petclaw/state.py:199-269 (DMDA ranges + globalToLocal), apps/euler/2d/shockbubble/
shockbubble.py:161-222, apps/shallow-sphere/shallow_4_Rossby_Haurwitz_wave.py:295-313 (pole boundary).
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import mp_fullsize_worker as W            # noqa: E402
from oracle import driver as D            # noqa: E402
from oracle import oracle as O            # noqa: E402


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch(case, nranks, expect_path, overlap=None, timeout=900, ahead=None):
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port), "PCL_HALO_TRANSPORT": "host", "PCL_FORCE_DEVICE": "0",
                    "TORCHELASTIC_RUN_ID": "fs%d" % port, "PCL_FS_EXPECT": expect_path})
        if overlap is not None:
            env["PCL_HALO_OVERLAP"] = str(overlap)
        if ahead is not None:
            env["PCL_EXCHANGE_AHEAD"] = str(ahead)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_fullsize_worker.py"), case],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    codes = [p.returncode for p in procs]
    assert codes == [0] * nranks, "exit codes %s\n%s" % (codes, "\n----\n".join(o[-2000:] for o in outs))
    assert "bit-identical to the serial run: True" in outs[0], outs[0][-2000:]
    return outs[0]


def serial_and_expect(case, tmp_path):
    """the serial run + the JSON of block hashes the workers compare with"""
    import pyclaw_amd as pyclaw
    from pyclaw_amd import parallel
    nx, ny, pg, _ = W.case_shape(case)
    q, rng, out = W.run_case(case, pyclaw)
    assert rng == [(0, nx), (0, ny)] and np.isfinite(q).all()
    px, py = (int(v) for v in pg.split("x"))
    blocks = {}
    for cy in range(py):
        for cx in range(px):
            (a, b), (c, d) = parallel.block_range(nx, px, cx), parallel.block_range(ny, py, cy)
            blocks["%d:%d,%d:%d" % (a, b, c, d)] = W.block_hash(q[:, a:b, c:d])
    path = os.path.join(str(tmp_path), "expect_%s.json" % case)
    with open(path, "w") as f:
        json.dump({"blocks": blocks, "status": out}, f)
    return q, out, path


def euler_window_replay(coracle, case, i0, j0, w, pad, steps):
    """the oracle's result for cells [i0, i0+w) x [j0, j0+w) of a synthetic Euler case after `steps` fixed-dt steps,
    from the initial cells of the window + `pad` cells of context on every side (periodic global indexing)"""
    nx, ny, _, _ = W.case_shape(case)
    ii = np.arange(i0 - pad, i0 + w + pad) % nx
    jj = np.arange(j0 - pad, j0 + w + pad) % ny
    q0 = W.synth_euler(ii, jj)
    dx, dy = (2.0 - 0.0) / float(nx), (2.0 * ny / nx - 0.0) / float(ny)
    aux = np.empty((1, len(ii), len(jj)), order="F")
    aux[0] = (0.0 + (jj + 0.5) * dy)[None, :]
    dt = W.fixed_dt(case)
    n = w + 2 * pad
    p = D.shockbubble_problem(mx=n, my=n, q=q0, aux=aux, d=(dx, dy), dim_split=case != "c4_unsplit", order_trans=2,
                              bc_lower=[D.OUTFLOW, D.OUTFLOW], bc_upper=[D.OUTFLOW, D.OUTFLOW], user_bc_lower=None,
                              cfl_max=1.0, cfl_desired=0.9, dt_initial=dt, dt_variable=False)
    D.run(p, coracle, steps * dt, 1)
    return p.q[:, pad:-pad, pad:-pad]


def wrap_take(q, i0, j0, w):
    nx, ny = q.shape[1:]
    return q[:, np.arange(i0, i0 + w) % nx][:, :, np.arange(j0, j0 + w) % ny]


@pytest.mark.parametrize("case", ["c4_periodic", "c4_layout", "c4_unsplit"])
def test_c4_blocks_decomposed_equals_serial_and_oracle_windows(case, coracle, tmp_path):
    nx, ny, pg, steps = W.case_shape(case)
    q, out, path = serial_and_expect(case, tmp_path)
    assert out["numsteps"] == steps and 0.05 < float(out["cflmax"]) < 1.0, out
    # windows across the block faces, at the four-block junction, across the periodic wrap and inside a block;
    # reach per step: 2 cells per pass (dim-split), 3 with the transverse terms
    w, pad = 24, (3 if case == "c4_unsplit" else 2) * steps + 2
    hx, hy = nx // 2, ny // 2
    wins = [(hx - w // 2, hy - w // 2), (hx - w // 2, hy // 2), (hx // 2, hy - w // 2), (-w // 2, -w // 2),
            (hx - w // 2, -w // 2), (nx - w // 2, hy - 5), (hx + 311, hy // 2 + 77), (hx - w + 1, hy - 1)]
    for (i0, j0) in wins:
        ref = euler_window_replay(coracle, case, i0, j0, w, pad, steps)
        got = wrap_take(q, i0, j0, w)
        assert np.array_equal(got, ref), (case, i0, j0, float(np.abs(got - ref).max()))
        assert not np.array_equal(ref, W.synth_euler(np.arange(i0, i0 + w) % nx, np.arange(j0, j0 + w) % ny))
    del q
    # Blocks that can all run the one-kernel step exchange in front of the step by default (and run its faster form);
    # PCL_EXCHANGE_AHEAD=2 asks for that step's exchange-ahead order (rim tiles first, the new halo behind them):
    # c4_periodic runs the default, c4_layout the exchange-ahead order; the unsplit step has none
    log = launch(case, 4, path, ahead=2 if case == "c4_layout" else None)
    assert ("exchange-ahead True" in log) == (case == "c4_layout"), log[-600:]


def test_c4_layout_sequential_exchange(tmp_path):
    """the C4 block shape once more with PCL_HALO_OVERLAP=0 (exchange in front of the step, one stream)"""
    q, out, path = serial_and_expect("c4_layout", tmp_path)
    del q
    log = launch("c4_layout", 4, path, overlap=0)
    assert "exchange-ahead False" in log


@pytest.mark.parametrize("ahead", [1, 2])
def test_c4_app_shockbubble_8192_decomposed_equals_serial(ahead, tmp_path):
    """the shock-bubble app itself on the C4 grid: inflow / reflecting / outflow sides on the edge blocks only,
    adaptive dt from the global Courant number (incl. the rejected first step), source term fused into the y pass;
    in the default order (exchange, then the step in its faster form) and in the one-kernel step's exchange-ahead order"""
    q, out, path = serial_and_expect("c4_app", tmp_path)
    assert out["numsteps"] >= 3, out
    # the post-shock inflow state has entered on the left, the bubble is still where it was
    assert q[1, 0, :].min() > 0.0 and q[0].min() < 0.2
    del q
    log = launch("c4_app", 4, path, ahead=ahead)
    assert "steps %d" % out["numsteps"] in log
    # (exchange-ahead incl. the rejected first step: back to the pre-step buffer's ghost frame)
    assert ("exchange-ahead True" in log) == (ahead == 2), log[-600:]


def sphere_window_replay(coracle, q0g, auxg, i0, j0, w, pad, dt, dx, dy):
    """SharpClaw WENO5 + SSP104 step of the sphere solver on a window of the global initial state (interior arrays
    q0g (4, nx, ny), auxg (16, nx, ny)), away from the poles"""
    sl = (slice(None), slice(i0 - pad, i0 + w + pad), slice(j0 - pad, j0 + w + pad))
    p = D.Problem(q=np.array(q0g[sl], order="F"), aux=np.array(auxg[sl], order="F"), d=(dx, dy),
                  rp=O.RP_SHALLOW_SPHERE_2D, rp_params=[11489.57219, dx, dy], mwaves=3, limiters=[1],
                  solver_type='sharpclaw', lim_type=2, mcapa=0, cfl_max=2.5, cfl_desired=2.45, dt_variable=False,
                  dt_initial=dt, bc_lower=[D.OUTFLOW, D.OUTFLOW], bc_upper=[D.OUTFLOW, D.OUTFLOW],
                  aux_bc_lower=[D.OUTFLOW, D.OUTFLOW], aux_bc_upper=[D.OUTFLOW, D.OUTFLOW])
    D.run(p, coracle, dt, 1)
    return p.q[:, pad:-pad, pad:-pad]


def test_c5_sphere_sharpclaw_2048x1024_as_four_blocks(coracle, tmp_path):
    from apps import shallow_sphere as S
    case = "c5_sphere"
    nx, ny, pg, steps = W.case_shape(case)
    q, out, path = serial_and_expect(case, tmp_path)
    assert out["numsteps"] == 1 and 0.2 < float(out["cflmax"]) < 2.5, out
    dx, dy = 4.0 / nx, 2.0 / ny
    dt = float(out["dt"])
    auxg = S.setaux(nx, ny, 3, -3.0, -1.0, dx, dy)[:, 3:-3, 3:-3]
    q0g = S.qinit(nx, ny, -3.0, -1.0, dx, dy)
    w, pad = 16, 32                      # ten stages of reach 3
    qy = ny // 4
    wins = [(nx // 2, qy - w // 2), (nx // 3, 2 * qy - w // 2), (nx - 200, 3 * qy - w // 2), (100, qy - 3),
            (nx // 2 - w // 2, ny // 2 + 40)]
    for (i0, j0) in wins:
        ref = sphere_window_replay(coracle, q0g, auxg, i0, j0, w, pad, dt, dx, dy)
        got = q[:, i0:i0 + w, j0:j0 + w]
        assert np.array_equal(got, ref), (i0, j0, float(np.abs(got - ref).max()))
        assert not np.array_equal(ref, q0g[:, i0:i0 + w, j0:j0 + w])
    del q
    launch(case, 4, path)
