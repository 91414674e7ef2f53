import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyclaw_amd import _lib as L
rng = np.random.default_rng(2)
mx, my = 90, 70
q0 = np.empty((5, mx + 6, my + 6), order="F")
rho = 0.5 + rng.random(q0.shape[1:]); u = rng.random(q0.shape[1:]) - 0.5; v = rng.random(q0.shape[1:]) - 0.5
q0[0], q0[1], q0[2] = rho, rho * u, rho * v
q0[3] = (0.5 + rng.random(q0.shape[1:])) / 0.4 + 0.5 * rho * (u * u + v * v)
q0[4] = rng.random(q0.shape[1:])
res = {}
for math in (0, 1):
    cfg = L.Config()
    cfg.ndim = 2; cfg.n[0], cfg.n[1] = mx, my; cfg.mbc = 3; cfg.meqn = cfg.mwaves = 5; cfg.rp = 11
    cfg.method[1] = 2; cfg.rp_params[0], cfg.rp_params[1] = 1.4, 0.4
    cfg.d[0], cfg.d[1] = 0.01, 0.012; cfg.kind = 1; cfg.lim_type = 2; cfg.math = math
    h = C.c_void_p()
    L.check(L.lib().pcl_create(C.byref(cfg), C.byref(h)))
    L.check(L.lib().pcl_put_q(h, L.d(q0), 1))
    cfl = C.c_double()
    L.check(L.lib().pcl_sharp_dq(h, 1e-3, C.cast(C.byref(cfl), L.dp)))
    L.check(L.lib().pcl_select(h, 3))
    dq = np.zeros_like(q0)
    L.check(L.lib().pcl_get_q(h, L.d(dq), 1))
    L.lib().pcl_destroy(h)
    res[math] = dq[:, 3:-3, 3:-3]
    print("math", math, "nan count", np.isnan(res[math]).sum(), "cfl", cfl.value)
n = np.argwhere(np.isnan(res[1]))
print(n[:20])
if len(n):
    m, i, j = n[0]
    print("q at", i, j, q0[:, i + 3, j + 3], "exact dq", res[0][:, i, j], "fast", res[1][:, i, j])
print("max rel diff where finite", np.nanmax(np.abs(res[1] - res[0])) / np.abs(res[0]).max())
