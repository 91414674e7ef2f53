"""
ctypes binding of libpyclaw_amd.so (include/pyclaw_amd.h).

This is the ONLY compute path of the package: there is no numpy/CPU fallback.  If the
shared library is missing, or no HIP device is present, the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCL_LIB_OVERRIDE") or os.path.join(_HERE, "libpyclaw_amd.so")   # override: A/B builds in tools/kbench.py

MAX_WAVES = 8
MAX_RP_PARAMS = 8

OK, EINVAL, ENODEVICE, EHIP, ECOMM, ESTATE = 0, -1, -2, -3, -4, -5

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


class Config(C.Structure):
    """struct pcl_config"""
    _fields_ = [
        ("ndim", C.c_int),
        ("n", C.c_int * 3),
        ("mbc", C.c_int),
        ("meqn", C.c_int),
        ("mwaves", C.c_int),
        ("maux", C.c_int),
        ("method", C.c_int * 7),
        ("mthlim", C.c_int * MAX_WAVES),
        ("fwave", C.c_int),
        ("rp", C.c_int),
        ("rp_params", C.c_double * MAX_RP_PARAMS),
        ("d", C.c_double * 3),
        ("device", C.c_int),
        ("math", C.c_int),
        ("kind", C.c_int),
        ("lim_type", C.c_int),
    ]


# name -> (restype, argtypes); every symbol declared in include/pyclaw_amd.h
PROTOTYPES = {
    "pcl_last_error": (C.c_char_p, []),
    "pcl_version": (C.c_int, []),
    "pcl_device_count": (C.c_int, []),
    "pcl_layer1_release": (None, []),
    "pcl_layer1_math": (C.c_int, [C.c_int]),
    "pcl_step1": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp,
                            C.c_double, C.c_double, ip, ip, dp]),
    "pcl_step1fw": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, dp,
                              C.c_double, C.c_double, ip, ip, dp]),
    "pcl_step2ds": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double, ip, ip, dp,
                              C.c_int]),
    "pcl_step2": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                            C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double, ip, ip, dp]),
    "pcl_step3ds": (C.c_int, [C.c_int, dp] + [C.c_int] * 7 + [dp, dp, dp] + [C.c_double] * 4 + [ip, ip, dp, C.c_int]),
    "pcl_step3": (C.c_int, [C.c_int, dp] + [C.c_int] * 7 + [dp, dp, dp] + [C.c_double] * 4 + [ip, ip, dp]),
    "pcl_sharp_module_mthlim": (C.c_int, [ip, C.c_int]),
    "pcl_sharp_module_char_decomp": (C.c_int, [C.c_int]),
    "pcl_sharp_flux1": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  dp, dp, dp, C.c_double, C.c_double, dp]),
    "pcl_sharp_flux2": (C.c_int, [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, dp, dp, dp, C.c_double, C.c_double, C.c_double, dp]),
    "pcl_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "pcl_destroy": (None, [C.c_void_p]),
    "pcl_put_q": (C.c_int, [C.c_void_p, dp, C.c_int]),
    "pcl_get_q": (C.c_int, [C.c_void_p, dp, C.c_int]),
    "pcl_put_aux": (C.c_int, [C.c_void_p, dp]),
    "pcl_bc": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pcl_bc_const": (C.c_int, [C.c_void_p, C.c_int, C.c_int, dp]),
    "pcl_bc_aux": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "pcl_get_strip": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]),
    "pcl_put_strip": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]),
    "pcl_put_aux_strip": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, dp]),
    "pcl_get_cells": (C.c_int, [C.c_void_p, C.c_int, ip, dp, dp]),
    "pcl_step_hyperbolic": (C.c_int, [C.c_void_p, C.c_double, dp]),
    "pcl_undo_step": (C.c_int, [C.c_void_p]),
    "pcl_bc_step": (C.c_int, [C.c_void_p, ip, dp, C.c_double, dp]),
    "pcl_sweep": (C.c_int, [C.c_void_p, C.c_int, C.c_double, dp]),
    "pcl_backup": (C.c_int, [C.c_void_p]),
    "pcl_restore": (C.c_int, [C.c_void_p]),
    "pcl_src": (C.c_int, [C.c_void_p, C.c_int, C.c_double, dp, C.c_int]),
    "pcl_fuse_source": (C.c_int, [C.c_void_p, C.c_int, dp, C.c_int]),
    "pcl_select": (C.c_int, [C.c_void_p, C.c_int]),
    "pcl_sharp_fuse_dq_src": (C.c_int, [C.c_void_p, C.c_int, dp, C.c_int]),
    "pcl_sharp_dq": (C.c_int, [C.c_void_p, C.c_double, dp]),
    "pcl_sharp_stage": (C.c_int, [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_double, C.c_double, dp]),
    "pcl_sharp_bc_dq": (C.c_int, [C.c_void_p, ip, dp, C.c_double, dp]),
    "pcl_sharp_bc_stage": (C.c_int, [C.c_void_p, ip, dp, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                     C.c_double, C.c_double, C.c_double, dp]),
    "pcl_rk_op": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                            C.c_double]),
    "pcl_sync": (C.c_int, [C.c_void_p]),
    "pcl_timer_start": (C.c_int, [C.c_void_p]),
    "pcl_timer_stop": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "pcl_kernel_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "pcl_kernel_timing_read": (C.c_int, [C.c_void_p, dp, C.POINTER(C.c_long)]),
    "pcl_step_form_stats": (C.c_int, [C.c_void_p, dp, C.POINTER(C.c_long), C.POINTER(C.c_long), C.POINTER(C.c_long)]),
    "pcl_step_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_long)]),
    "pcl_comm_unique_id": (C.c_int, [C.c_char_p]),
    "pcl_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_char_p, ip]),
    "pcl_comm_check": (C.c_int, [C.c_int, C.c_int, ip]),
    "pcl_comm_init_host": (C.c_int, [C.c_void_p, C.c_int, C.c_int, ip, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pcl_halo_can_overlap": (C.c_int, [C.c_void_p, ip]),
    "pcl_halo_exchange_ahead": (C.c_int, [C.c_void_p, C.c_int]),
    "pcl_halo_exchange": (C.c_int, [C.c_void_p]),
    "pcl_halo_exchange_aux": (C.c_int, [C.c_void_p]),
    "pcl_allreduce_max": (C.c_int, [C.c_void_p, dp]),
    "pcl_halo_region": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, ip]),
    "pcl_debug_wave_shift": (C.c_int, [dp, dp, dp]),
}

_lib = None


class PclError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libpyclaw_amd error %d: %s" % (code, msg))
        self.code = code


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libpyclaw_amd.so is not built (%s): run `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C pyclaw_amd/csrc`.  pyclaw_amd has no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            try:
                f = getattr(L, name)   # AttributeError if the ABI lacks a declared symbol
            except AttributeError:
                if os.environ.get("PCL_LIB_OVERRIDE"):     # an older A/B build (tools/kbench.py) may predate a symbol
                    continue
                raise
            f.restype = res
            f.argtypes = args
        _lib = L
        # the cached device handle of the f2py-shaped calls is freed while the HIP runtime is still alive (the
        # library itself makes no HIP call from a static destructor)
        import atexit
        atexit.register(L.pcl_layer1_release)
    return _lib


def check(rc):
    if rc != 0:
        raise PclError(rc, lib().pcl_last_error().decode("utf-8", "replace"))


def d(a):
    return a.ctypes.data_as(dp)


def i(a):
    return a.ctypes.data_as(ip)


def fortran64(a):
    """float64 Fortran-contiguous view/copy of a."""
    a = np.asarray(a, dtype=np.float64)
    if not a.flags.f_contiguous:
        a = np.asfortranarray(a)
    return a
