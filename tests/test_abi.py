"""
CPU: the C-ABI shared library loads without a GPU and exports every symbol that
include/pyclaw_amd.h declares; compute entry points fail loudly (no CPU fallback).
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "pyclaw_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pcl_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_expected_surface():
    syms = header_symbols()
    for must in ["pcl_step1", "pcl_step2ds", "pcl_step2", "pcl_create", "pcl_step_hyperbolic",
                 "pcl_halo_exchange", "pcl_allreduce_max", "pcl_put_q", "pcl_get_q", "pcl_bc"]:
        assert must in syms


def test_library_exports_every_declared_symbol():
    from pyclaw_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = C.CDLL(_lib.LIB_PATH)
    missing = [s for s in header_symbols() if not hasattr(L, s)]
    assert not missing, missing
    # and the Python binding table covers the same set
    assert sorted(_lib.PROTOTYPES) == header_symbols()


def test_struct_layout_matches_header():
    from pyclaw_amd import _lib
    # ndim + n[3] + mbc + meqn,mwaves,maux + method[7] + mthlim[8] + fwave + rp = 25 ints (+4 pad),
    # then 8 + 3 doubles, then device, math, kind, lim_type
    assert _lib.Config.rp_params.offset == 104
    assert C.sizeof(_lib.Config) == 104 + 11 * 8 + 4 * 4


def test_no_cpu_fallback():
    """Without a HIP device every compute entry point must fail (this container has no GPU;
    on the GPU box the test is skipped)."""
    from pyclaw_amd import _lib
    L = _lib.lib()
    if L.pcl_device_count() > 0:
        pytest.skip("a GPU is present")
    cfg = _lib.Config()
    cfg.ndim = 1
    cfg.n[0] = 10
    cfg.mbc = 2
    cfg.meqn = 1
    cfg.mwaves = 1
    cfg.rp = 1
    cfg.mthlim[0] = 4
    cfg.method[1] = 2
    cfg.d[0] = 0.1
    h = C.c_void_p()
    rc = L.pcl_create(C.byref(cfg), C.byref(h))
    assert rc == _lib.ENODEVICE
    assert b"no HIP device" in L.pcl_last_error()
    q = np.zeros((1, 14), order="F")
    cfl = C.c_double()
    method = np.array([1, 2, 0, 0, 0, 0, 0], dtype=np.int32)
    mth = np.array([4], dtype=np.int32)
    par = np.zeros(8)
    rc = L.pcl_step1(1, _lib.d(par), 1, 1, 0, 2, 10, _lib.d(q), None, 0.1, 0.01, _lib.i(method), _lib.i(mth),
                     C.cast(C.byref(cfl), _lib.dp))
    assert rc == _lib.ENODEVICE


def test_halo_region_geometry():
    """pcl_halo_region is pure host code: send strips are interior cells, fill strips are ghosts,
    and what is sent towards d has the shape of what the neighbour fills from opposite(d)."""
    from pyclaw_amd import _lib
    L = _lib.lib()
    I, J, g = 15, 11, 2
    opp = [1, 0, 3, 2, 7, 6, 5, 4]
    o = np.zeros(4, dtype=np.int32)
    covered = np.zeros((I, J), dtype=int)
    for d in range(8):
        _lib.check(L.pcl_halo_region(d, 1, I, J, g, _lib.i(o)))
        si0, sj0, sni, snj = o
        assert g <= si0 and si0 + sni <= I - g and g <= sj0 and sj0 + snj <= J - g
        _lib.check(L.pcl_halo_region(opp[d], 0, I, J, g, _lib.i(o)))
        fi0, fj0, fni, fnj = o
        assert (sni, snj) == (fni, fnj)
        _lib.check(L.pcl_halo_region(d, 0, I, J, g, _lib.i(o)))
        covered[o[0]:o[0] + o[2], o[1]:o[1] + o[3]] += 1
    # the 8 fill strips tile the ghost frame exactly once and never touch the interior
    frame = np.ones((I, J), dtype=int)
    frame[g:-g, g:-g] = 0
    assert np.array_equal(covered, frame)
