"""
CPU: the window-replay devices of tests/test_gpu_c4c5.py checked against full-grid oracle replays on the small twins
of the C4 / C5 cases (PCL_FS_SMALL geometry): a window with `pad` cells of context, stepped by the oracle with
throw-away boundary values, must reproduce the same cells of the oracle's run over the whole periodic grid -- i.e. the
context is sufficient, the periodic indexing and the aux rows are right -- and the synthetic state is identical
whichever block computes a cell.
"""
import os

import numpy as np
import pytest

import mp_fullsize_worker as W            # noqa: E402
import test_gpu_c4c5 as T                # noqa: E402
from oracle import driver as D            # noqa: E402


@pytest.fixture(autouse=True)
def small_geometry(monkeypatch):
    monkeypatch.setenv("PCL_FS_SMALL", "1")


def test_synthetic_state_is_block_independent():
    full = W.synth_euler(np.arange(256), np.arange(192))
    assert np.array_equal(full[:, 128:, 96:], W.synth_euler(np.arange(128, 256), np.arange(96, 192)))
    assert np.array_equal(full[:, 3:77, 150:], W.synth_euler(np.arange(3, 77), np.arange(150, 192)))
    assert full[0].min() > 0.25 and np.unique(full[0]).size > 500


@pytest.mark.parametrize("case", ["c4_periodic", "c4_unsplit"])
def test_euler_windows_equal_the_full_grid_oracle(case, coracle):
    nx, ny, _, steps = W.case_shape(case)
    dx, dy = 2.0 / float(nx), (2.0 * ny / nx) / float(ny)
    q0 = W.synth_euler(np.arange(nx), np.arange(ny))
    aux = np.empty((1, nx, ny), order="F")
    aux[0] = ((np.arange(ny) + 0.5) * dy)[None, :]
    dt = W.fixed_dt(case)
    p = D.shockbubble_problem(mx=nx, my=ny, q=q0.copy("F"), aux=aux, d=(dx, dy), dim_split=case != "c4_unsplit",
                              order_trans=2, bc_lower=[D.PERIODIC] * 2, bc_upper=[D.PERIODIC] * 2, user_bc_lower=None,
                              cfl_max=1.0, cfl_desired=0.9, dt_initial=dt, dt_variable=False)
    st = D.run(p, coracle, steps * dt, 1)
    assert st[-1]["numsteps"] == steps and 0.05 < st[-1]["cflmax"] < 1.0
    w, pad = 24, (3 if case == "c4_unsplit" else 2) * steps + 2
    for (i0, j0) in [(nx // 2 - 12, ny // 2 - 12), (-12, -12), (nx - 12, 40), (100, ny - 5), (7, 9)]:
        ref = T.euler_window_replay(coracle, case, i0, j0, w, pad, steps)
        assert np.array_equal(ref, T.wrap_take(p.q, i0, j0, w)), (i0, j0)
    if case == "c4_periodic":     # less context is NOT enough: the check can tell a wrong ghost cell from a right one
        short = T.euler_window_replay(coracle, case, nx // 2 - 12, ny // 2 - 12, w, pad - 3, steps)
        assert not np.array_equal(short, T.wrap_take(p.q, nx // 2 - 12, ny // 2 - 12, w))


def test_sphere_sharpclaw_window_equals_the_full_grid_oracle(coracle):
    nx, ny = 128, 96
    dx, dy = 4.0 / nx, 2.0 / ny
    p = D.shallow_sphere_problem(coracle, mx=nx, my=ny, solver_type='sharpclaw')
    q0 = p.q.copy("F")
    auxg = p.aux.copy("F")
    dt = 0.4 * min(dx, dy) / 4.0
    p.dt_variable, p.dt_initial = False, dt
    D.run(p, coracle, dt, 1)
    w, pad = 16, 32
    for (i0, j0) in [(40, 40), (nx - w - pad, ny - w - pad)]:
        ref = T.sphere_window_replay(coracle, q0, auxg, i0, j0, w, pad, dt, dx, dy)
        assert np.array_equal(ref, p.q[:, i0:i0 + w, j0:j0 + w]), (i0, j0)
