"""
oracle/oracle.py -- ctypes access to the parity oracle.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; nothing under pyclaw_amd/ does.

Two back ends with the same Python call surface:

* ``COracle``   -- this repo's C restatement (oracle/classic_oracle.c -> liboracle.so)
* ``RefEuler2D``-- the REFERENCE's own Fortran (src/fortran/2d/classic/{step2ds,step2,flux2}.f,
  1d/classic/{limiter,philim}.f, development/rp_approaches/rp{n,t}2_euler_5wave_rec_loc.f)
  compiled unchanged by oracle/Makefile with flang into oracle/_ref/.  f2py is not
  used: every scalar goes by reference, INTEGER*4 arrays as int32, arrays F-ordered.

Arrays follow the reference's f2py call surface (clawpack.py:538-552):
qold/qnew/aux are Fortran-ordered float64 ``(meqn, mx+2mbc, my+2mbc)``.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

RP_ADVECTION_1D = 1
RP_ACOUSTICS_1D = 2
RP_BURGERS_1D = 3
RP_EULER_1D = 4
RP_SHALLOW_1D = 5
RP_ADVECTION_COLOR_1D = 6
RP_ACOUSTICS_2D = 10
RP_ADVECTION_2D = 12
RP_SHALLOW_2D = 13
RP_VC_ACOUSTICS_2D = 14
RP_VC_ADVECTION_2D = 15
RP_EULER5_2D = 11
RP_SHALLOW_SPHERE_2D = 16
RP_ELASTICITY_FWAVE_1D = 7
RP_PSYSTEM_FWAVE_2D = 17
RP_VC_ACOUSTICS_3D = 20

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    a = np.asarray(a, dtype=np.float64)
    return a if a.flags.f_contiguous else np.asfortranarray(a)


class COracle:
    """C restatement (port).  See classic_oracle.c for the reference file:line map."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle not built: run `make -C oracle` (or __graft_entry__.build())")
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_step2ds.restype = C.c_int
        L.orc_step2ds.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double,
                                  C.c_double, _ip, _ip, _dp, C.c_int]
        L.orc_step2.restype = C.c_int
        L.orc_step2.argtypes = L.orc_step2ds.argtypes[:-1]
        L.orc_step3ds.restype = C.c_int
        L.orc_step3ds.argtypes = [C.c_int] * 9 + [_dp, _dp, _dp] + [C.c_double] * 4 + [_ip, _ip, _dp, C.c_int]
        L.orc_step3.restype = C.c_int
        L.orc_step3.argtypes = L.orc_step3ds.argtypes[:-1]
        L.orc_step1.restype = C.c_int
        L.orc_step1.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                                C.c_double, C.c_double, _ip, _ip, _dp]
        L.orc_step1fw.restype = C.c_int
        L.orc_step1fw.argtypes = L.orc_step1.argtypes
        L.orc_rpn2.restype = C.c_int
        L.orc_rpn2.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp,
                               _dp, _dp, _dp]
        L.orc_rpt2.restype = C.c_int
        L.orc_rpt2.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_limiter.restype = None
        L.orc_limiter.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _ip]
        L.orc_philim.restype = C.c_double
        L.orc_philim.argtypes = [C.c_double, C.c_double, C.c_int]
        L.orc_sharp_flux2.restype = C.c_int
        L.orc_sharp_flux2.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double, _dp]
        L.orc_sharp_flux1.restype = C.c_int
        L.orc_sharp_flux1.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp]
        L.orc_weno5.restype = None
        L.orc_weno5.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_weno_k.restype = None
        L.orc_weno_k.argtypes = [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]
        L.orc_sharp_set_weno_order.restype = None
        L.orc_sharp_set_weno_order.argtypes = [C.c_int]
        L.orc_sharp_set_mthlim.restype = None
        L.orc_sharp_set_mthlim.argtypes = [_ip, C.c_int]
        L.orc_sphere_qcor.restype = None
        L.orc_sphere_qcor.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.orc_set_qcor.restype = None
        L.orc_set_qcor.argtypes = [C.c_int]
        L.orc_sphere_mapc2p.restype = None
        L.orc_sphere_mapc2p.argtypes = [C.c_double, C.c_double, _dp, _dp, _dp, C.c_double]
        L.orc_sphere_setaux.restype = C.c_int
        L.orc_sphere_setaux.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + [_dp, C.c_double]
        L.orc_sphere_qinit.restype = C.c_int
        L.orc_sphere_qinit.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + [_dp, C.c_double]
        L.orc_sphere_src2.restype = C.c_int
        L.orc_sphere_src2.argtypes = [C.c_int, C.c_int, C.c_int] + [C.c_double] * 4 + [_dp, C.c_int, _dp, C.c_double,
                                                                                      C.c_double]

    # -- f2py-shaped entry points -------------------------------------------------
    def step2ds(self, rp, par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method, mthlim, ids,
                fwave=False):
        """classic2.step2ds(...) -> (qnew, cfl); qnew updated in place (may alias qold)."""
        meqn = qnew.shape[0]
        par = np.ascontiguousarray(par, dtype=np.float64)
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        maux = int(method[6])
        auxp = _d(aux) if (aux is not None and maux > 0) else _d(np.zeros(1))
        cfl = C.c_double(0.0)
        assert qnew.flags.f_contiguous and qold.flags.f_contiguous
        rc = self.lib.orc_step2ds(rp, _d(par), int(fwave), maxm, meqn, len(mthlim), maux, mbc, mx, my,
                                  _d(qold), _d(qnew), auxp, dx, dy, dt, _i(method), _i(mthlim),
                                  C.byref(cfl), ids)
        if rc:
            raise RuntimeError("oracle: unknown Riemann solver id %d" % rp)
        return qnew, cfl.value

    def step3ds(self, rp, maxm, mbc, mx, my, mz, qold, qnew, aux, dx, dy, dz, dt, method, mthlim, idir):
        """classic3.step3ds(...) -> (qnew, cfl) (clawpack.py:678-688); qnew in place, may alias qold."""
        meqn = qnew.shape[0]
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        maux = int(method[6])
        auxp = _d(aux) if (aux is not None and maux > 0) else _d(np.zeros(1))
        cfl = C.c_double(0.0)
        assert qnew.flags.f_contiguous and qold.flags.f_contiguous
        rc = self.lib.orc_step3ds(rp, maxm, meqn, len(mthlim), maux, mbc, mx, my, mz, _d(qold), _d(qnew), auxp,
                                  dx, dy, dz, dt, _i(method), _i(mthlim), C.byref(cfl), idir)
        if rc:
            raise RuntimeError("oracle step3ds: rc=%d (unknown Riemann solver / unsplit 3-D not restated)" % rc)
        return qnew, cfl.value

    def step3(self, rp, maxm, mbc, mx, my, mz, qold, qnew, aux, dx, dy, dz, dt, method, mthlim):
        """classic3.step3(...) -> (qnew, cfl) (clawpack.py:690-696): the unsplit 3-D algorithm (step3.f + flux3.f)"""
        meqn = qnew.shape[0]
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        maux = int(method[6])
        auxp = _d(aux) if (aux is not None and maux > 0) else _d(np.zeros(1))
        cfl = C.c_double(0.0)
        assert qnew.flags.f_contiguous and qold.flags.f_contiguous
        rc = self.lib.orc_step3(rp, maxm, meqn, len(mthlim), maux, mbc, mx, my, mz, _d(qold), _d(qnew), auxp,
                                dx, dy, dz, dt, _i(method), _i(mthlim), C.byref(cfl))
        if rc:
            raise RuntimeError("oracle step3: rc=%d" % rc)
        return qnew, cfl.value

    def step2(self, rp, par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method, mthlim,
              fwave=False):
        meqn = qnew.shape[0]
        par = np.ascontiguousarray(par, dtype=np.float64)
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        maux = int(method[6])
        auxp = _d(aux) if (aux is not None and maux > 0) else _d(np.zeros(1))
        cfl = C.c_double(0.0)
        assert qnew.flags.f_contiguous and qold.flags.f_contiguous
        rc = self.lib.orc_step2(rp, _d(par), int(fwave), maxm, meqn, len(mthlim), maux, mbc, mx, my,
                                _d(qold), _d(qnew), auxp, dx, dy, dt, _i(method), _i(mthlim),
                                C.byref(cfl))
        if rc:
            raise RuntimeError("oracle: unknown Riemann solver id %d" % rp)
        return qnew, cfl.value

    def step1(self, rp, par, mbc, mx, q, aux, dx, dt, method, mthlim, fwave=False):
        """classic1.step1 (step1.f) or, fwave=True, classic1fw.step1 (step1fw.f)"""
        meqn = q.shape[0]
        par = np.ascontiguousarray(par, dtype=np.float64)
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        maux = int(method[6])
        auxp = _d(aux) if (aux is not None and maux > 0) else _d(np.zeros(1))
        cfl = C.c_double(0.0)
        assert q.flags.f_contiguous
        fn = self.lib.orc_step1fw if fwave else self.lib.orc_step1
        rc = fn(rp, _d(par), meqn, len(mthlim), maux, mbc, mx, _d(q), auxp, dx, dt, _i(method), _i(mthlim), C.byref(cfl))
        if rc:
            raise RuntimeError("oracle: unknown Riemann solver id %d" % rp)
        return q, cfl.value

    # -- SharpClaw (sharpclaw_oracle.c) ------------------------------------------------
    def set_sharp_mthlim(self, mthlim):
        """clawparams.mthlim (sharpclaw.py:268): used by lim_type=1 (tvd2), indexed by component"""
        m = np.ascontiguousarray(mthlim, dtype=np.int32)
        self.lib.orc_sharp_set_mthlim(_i(m), len(m))

    def set_char_decomp(self, char_decomp, fwave=False):
        """clawparams.char_decomp / fwave (sharpclaw.py:262-266): 1 = wave-based reconstruction in the 1-D flux1"""
        self.lib.orc_sharp_set_char_decomp(int(char_decomp), int(bool(fwave)))

    def recon_wave(self, kind, q, wave, s, mthlim):
        """tvd2_wave (kind 1) / weno5_wave (2) / weno5_fwave (3) of 1d/sharpclaw/reconstruct.f90 on given arrays
        q (meqn, n), wave (meqn, mwaves, n), s (mwaves, n); indices outside the arrays read as 0 -> (ql, qr)"""
        meqn, n = q.shape
        mwaves = wave.shape[1]
        q = np.asfortranarray(q, dtype=np.float64)
        w = np.array(wave, dtype=np.float64, order="F")
        s = np.asfortranarray(s, dtype=np.float64)
        ql = np.zeros((meqn, n), order="F")
        qr = np.zeros((meqn, n), order="F")
        m = np.ascontiguousarray(mthlim, dtype=np.int32)
        assert self.lib.orc_recon_wave(int(kind), meqn, mwaves, n, _d(q), _d(w), _d(s), _i(m), _d(ql), _d(qr)) == 0
        return ql, qr

    def sharp_flux2(self, rp, par, lim_type, mwaves, mcapa, mbc, mx, my, q, aux, dx, dy, dt):
        """sharpclaw2.flux2(q,aux,dt,t,mbc,maxm,mx,my) -> (dq,cfl)  (sharpclaw.py:558)"""
        meqn = q.shape[0]
        par = np.ascontiguousarray(par, dtype=np.float64)
        maux = 0 if aux is None else aux.shape[0]
        dq = np.zeros(q.shape, order="F")
        cfl = C.c_double(0.0)
        assert q.flags.f_contiguous
        rc = self.lib.orc_sharp_flux2(rp, _d(par), lim_type, meqn, mwaves, maux, mcapa, mbc, mx, my, _d(q),
                                      _d(dq), _d(aux) if maux else None, dx, dy, dt, C.byref(cfl))
        if rc:
            raise RuntimeError("oracle sharpclaw: rc=%d" % rc)
        return dq, cfl.value

    def sharp_flux1(self, rp, par, lim_type, mwaves, mcapa, mbc, mx, q, aux, dx, dt):
        """sharpclaw1.flux1(q,aux,dt,t,ixy,mx,mbc,maxnx) -> (dq,cfl)  (sharpclaw.py:385)"""
        meqn = q.shape[0]
        par = np.ascontiguousarray(par, dtype=np.float64)
        maux = 0 if aux is None else aux.shape[0]
        dq = np.zeros(q.shape, order="F")
        cfl = C.c_double(0.0)
        rc = self.lib.orc_sharp_flux1(rp, _d(par), lim_type, meqn, mwaves, maux, mcapa, mbc, mx, _d(_f64(q)),
                                      _d(dq), _d(aux) if maux else None, dx, dt, C.byref(cfl))
        if rc:
            raise RuntimeError("oracle sharpclaw: rc=%d" % rc)
        return dq, cfl.value

    def set_weno_order(self, order):
        """clawparams.weno_order (sharpclaw.py:263): 5 .. 17, odd; mbc must be (order+1)/2"""
        self.lib.orc_sharp_set_weno_order(int(order))

    def weno_k(self, order, q):
        meqn, n = q.shape
        ql = np.zeros((meqn, n), order="F")
        qr = np.zeros((meqn, n), order="F")
        self.lib.orc_weno_k(int(order), meqn, n, _d(_f64(q)), _d(ql), _d(qr))
        return ql, qr

    def weno5(self, variant, mbc, q):
        meqn, n = q.shape
        ql = np.zeros((meqn, n), order="F")
        qr = np.zeros((meqn, n), order="F")
        self.lib.orc_weno5(variant, meqn, n, mbc, _d(_f64(q)), _d(ql), _d(qr))
        return ql, qr

    # -- shallow water on the sphere: the app's own Fortran restated (sphere_oracle.c) -----------
    def set_qcor(self, on):
        """step2qcor.f in place of step2.f (test/shallow_sphere/Makefile:16)"""
        self.lib.orc_set_qcor(int(bool(on)))

    def qcor(self, ixy, i, aux1d, q1d, mbc, g, dx, dy):
        qc = np.zeros(4)
        par = np.array([g, dx, dy])
        self.lib.orc_sphere_qcor(ixy, i, _d(_f64(aux1d)), _d(_f64(q1d)), q1d.shape[0], mbc, _d(par), _d(qc))
        return qc

    def sphere_setaux(self, mbc, mx, my, xlower, ylower, dx, dy, Rsphere=1.0):
        aux = np.zeros((16, mx + 2 * mbc, my + 2 * mbc), order="F")
        assert self.lib.orc_sphere_setaux(mbc, mx, my, xlower, ylower, dx, dy, _d(aux), Rsphere) == 0
        return aux

    def sphere_qinit(self, mbc, mx, my, xlower, ylower, dx, dy, Rsphere=1.0):
        q = np.zeros((4, mx + 2 * mbc, my + 2 * mbc), order="F")
        assert self.lib.orc_sphere_qinit(mbc, mx, my, xlower, ylower, dx, dy, _d(q), Rsphere) == 0
        return q

    def sphere_src2(self, q, aux, xlower, ylower, dx, dy, dt, Rsphere=1.0):
        """problem.src2(mx,my,mbc,xlower,ylower,dx,dy,q,aux,t,dt,Rsphere) on the interior arrays; q in place"""
        assert q.flags.f_contiguous and aux.flags.f_contiguous
        meqn, mx, my = q.shape
        assert self.lib.orc_sphere_src2(meqn, mx, my, xlower, ylower, dx, dy, _d(q), aux.shape[0], _d(aux), dt,
                                        Rsphere) == 0
        return q

    # -- slice-level pieces -------------------------------------------------------
    def rpn2(self, rp, par, ixy, mwaves, mbc, mx, q1d):
        meqn = q1d.shape[0]
        n = mx + 2 * mbc
        par = np.ascontiguousarray(par, dtype=np.float64)
        wave = np.zeros((meqn, mwaves, n), order="F")
        s = np.zeros((mwaves, n), order="F")
        amdq = np.zeros((meqn, n), order="F")
        apdq = np.zeros((meqn, n), order="F")
        rc = self.lib.orc_rpn2(rp, _d(par), ixy, meqn, mwaves, mbc, mx, _d(_f64(q1d)), _d(wave), _d(s),
                               _d(amdq), _d(apdq))
        assert rc == 0
        return wave, s, amdq, apdq

    def rpt2(self, rp, par, ixy, mbc, mx, q1d, asdq):
        meqn = q1d.shape[0]
        n = mx + 2 * mbc
        par = np.ascontiguousarray(par, dtype=np.float64)
        bm = np.zeros((meqn, n), order="F")
        bp = np.zeros((meqn, n), order="F")
        rc = self.lib.orc_rpt2(rp, _d(par), ixy, meqn, mbc, mx, _d(_f64(q1d)), _d(_f64(asdq)), _d(bm),
                               _d(bp))
        assert rc == 0
        return bm, bp

    def limiter(self, mbc, mx, wave, s, mthlim):
        meqn, mwaves, _ = wave.shape
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        wave = np.array(wave, order="F", dtype=np.float64)
        self.lib.orc_limiter(meqn, mwaves, mbc, mx, _d(wave), _d(_f64(s)), _i(mthlim))
        return wave

    def philim(self, a, b, meth):
        return self.lib.orc_philim(a, b, meth)


class RefEuler2D:
    """The reference Fortran itself (flang build under oracle/_ref/), Euler 5-wave only.

    Call order of the 28 by-reference arguments follows step2ds.f:2-5.
    """

    rp = RP_EULER5_2D

    class _CParam(C.Structure):
        _fields_ = [("gamma", C.c_double), ("gamma1", C.c_double)]

    def __init__(self, path=None, fwave=False):
        """fwave=True: the classic2fw link (flux2fw.f in place of flux2.f, oracle/Makefile: libref_euler2d_fw.so)"""
        path = path or os.path.join(_HERE, "_ref", "libref_euler2d_fw.so" if fwave else "libref_euler2d.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.fwave = bool(fwave)
        self.cparam = self._CParam.in_dll(self.lib, "cparam_")

    @staticmethod
    def available(fwave=False):
        return os.path.exists(os.path.join(_HERE, "_ref", "libref_euler2d_fw.so" if fwave else "libref_euler2d.so"))

    def _call(self, name, par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method, mthlim, ids):
        self.cparam.gamma, self.cparam.gamma1 = float(par[0]), float(par[1])
        meqn = qnew.shape[0]
        method = np.ascontiguousarray(method, dtype=np.int32)
        mthlim = np.ascontiguousarray(mthlim, dtype=np.int32)
        mwaves = len(mthlim)
        maux_decl = max(int(method[6]), 1)
        n = maxm + 2 * mbc
        qadd = np.zeros((meqn, n), order="F")
        fadd = np.zeros((meqn, n), order="F")
        gadd = np.zeros((meqn, 2, n), order="F")
        q1d = np.zeros((meqn, n), order="F")
        dtdx1d = np.zeros(n)
        dtdy1d = np.zeros(n)
        aux1 = np.zeros((maux_decl, n), order="F")
        aux2 = np.zeros((maux_decl, n), order="F")
        aux3 = np.zeros((maux_decl, n), order="F")
        mwork = n * (5 * meqn + mwaves + meqn * mwaves)
        work = np.zeros(mwork)
        if aux is None:
            aux = np.zeros((1,) + qnew.shape[1:], order="F")
        ci = lambda v: C.byref(C.c_int(v))
        cd = lambda v: C.byref(C.c_double(v))
        cfl = C.c_double(0.0)
        args = [ci(maxm), ci(meqn), ci(mwaves), ci(maux_decl), ci(mbc), ci(mx), ci(my),
                _d(qold), _d(qnew), _d(aux), cd(dx), cd(dy), cd(dt), _i(method), _i(mthlim),
                C.byref(cfl), _d(qadd), _d(fadd), _d(gadd), _d(q1d), _d(dtdx1d), _d(dtdy1d),
                _d(aux1), _d(aux2), _d(aux3), _d(work), ci(mwork)]
        if ids is not None:
            args.append(ci(ids))
        getattr(self.lib, name)(*args)
        return qnew, cfl.value

    def step2ds(self, rp, par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method, mthlim, ids,
                fwave=False):
        assert rp == RP_EULER5_2D and bool(fwave) == self.fwave
        return self._call("step2ds_", par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method,
                          mthlim, ids)

    def step2(self, rp, par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method, mthlim,
              fwave=False):
        assert rp == RP_EULER5_2D and bool(fwave) == self.fwave
        return self._call("step2_", par, maxm, mbc, mx, my, qold, qnew, aux, dx, dy, dt, method,
                          mthlim, None)


class RefSharp2DEuler:
    """The reference's SharpClaw Fortran (2d/sharpclaw/flux2.f90, flux1.f90, 1d/sharpclaw/weno.f90,
    reconstruct.f90 + vendored Euler rpn2), flang-built, driven through oracle/ref_sharpclaw_shim.f90.
    flux2_ takes 14 by-reference arguments in the order of flux2.f90:2."""

    class _CParam(C.Structure):
        _fields_ = [("gamma", C.c_double), ("gamma1", C.c_double)]

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "libref_sharpclaw2d_euler.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.cparam = self._CParam.in_dll(self.lib, "cparam_")

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libref_sharpclaw2d_euler.so"))

    def recon_wave(self, kind, q, wave, s, mthlim, pad=4):
        """the reference's own tvd2_wave / weno5_wave / weno5_fwave (oracle/ref_sharpclaw_shim.f90: sc_recon_wave).  The
        Fortran reads up to two entries outside its arrays: they are handed views into zero-padded buffers, so those
        reads see 0 (what the C restatement assumes).  Call before any sharp_flux2 of the process (shim comment)."""
        meqn, n = q.shape
        mwaves = wave.shape[1]
        N = n + 2 * pad
        qb = np.zeros((meqn, N), order="F"); qb[:, pad:pad + n] = q
        wb = np.zeros((meqn, mwaves, N), order="F"); wb[:, :, pad:pad + n] = wave
        sb = np.ones((mwaves, N), order="F"); sb[:, pad:pad + n] = s
        qlb = np.zeros((meqn, N), order="F"); qrb = np.zeros((meqn, N), order="F")
        m = np.ascontiguousarray(mthlim, dtype=np.int32)
        v = lambda a, k: a[(slice(None),) * k + (slice(pad, pad + n),)]
        qv, wv, sv, qlv, qrv = v(qb, 1), v(wb, 2), v(sb, 1), v(qlb, 1), v(qrb, 1)
        assert qv.flags.f_contiguous and wv.flags.f_contiguous and sv.flags.f_contiguous
        self.lib.sc_recon_wave(C.c_int(kind), C.c_int(meqn), C.c_int(mwaves), C.c_int(n), _d(qv), _d(qlv), _d(qrv), _d(wv),
                               _d(sv), _i(m))
        return np.array(qlv, order="F"), np.array(qrv, order="F")

    def sharp_flux2(self, rp, par, lim_type, mwaves, mcapa, mbc, mx, my, q, aux, dx, dy, dt, mthlim=None,
                    weno_order=5):
        assert rp == RP_EULER5_2D and mcapa == 0
        self.cparam.gamma, self.cparam.gamma1 = float(par[0]), float(par[1])
        meqn = q.shape[0]
        maxnx = max(mx, my) + 2 * mbc                      # sharpclaw.py:280
        dxs = np.array([dx, dy])
        mth = np.array(mthlim if mthlim is not None else [1] * mwaves, dtype=np.int32)
        self.lib.sc_setup(C.c_int(2), C.c_int(meqn), C.c_int(mwaves), C.c_int(mbc), C.c_int(maxnx),
                          C.c_int(lim_type), C.c_int(weno_order), C.c_int(0), C.c_int(0), _d(dxs), _i(mth))
        dq = np.zeros(q.shape, order="F")
        q1d = np.zeros((meqn, maxnx + 2 * mbc), order="F")
        dq1d = np.zeros((meqn, maxnx + 2 * mbc), order="F")
        auxd = np.zeros((1,) + q.shape[1:], order="F")
        cfl = C.c_double()
        ci = lambda v: C.byref(C.c_int(v))
        self.lib.flux2_(_d(q), _d(dq), _d(q1d), _d(dq1d), _d(auxd), C.byref(C.c_double(dt)), C.byref(cfl),
                        C.byref(C.c_double(0.0)), ci(0), ci(meqn), ci(mbc), ci(max(mx, my)), ci(mx), ci(my))
        return dq, cfl.value


class RefSphereProblem:
    """The shallow-sphere application's own Fortran (test/shallow_sphere/{mapc2p,setaux,qinit,src2,qcor}.f), flang-built
    by oracle/Makefile into oracle/_ref/libref_sphere_problem.so: what the reference's Makefile calls problem.so.
    By-reference arguments in the order of the Fortran statements (setaux.f:2-3, qinit.f:2-3, src2.f:2-3)."""

    def __init__(self, path=None):
        path = path or os.path.join(_HERE, "_ref", "libref_sphere_problem.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)

    @staticmethod
    def available():
        return os.path.exists(os.path.join(_HERE, "_ref", "libref_sphere_problem.so"))

    def sphere_setaux(self, mbc, mx, my, xlower, ylower, dx, dy, Rsphere=1.0):
        aux = np.zeros((16, mx + 2 * mbc, my + 2 * mbc), order="F")
        ci = lambda v: C.byref(C.c_int(v))
        cd = lambda v: C.byref(C.c_double(v))
        self.lib.setaux_(ci(mx), ci(my), ci(mbc), ci(mx), ci(my), cd(xlower), cd(ylower), cd(dx), cd(dy), ci(16),
                         _d(aux), cd(Rsphere))
        return aux

    def sphere_qinit(self, mbc, mx, my, xlower, ylower, dx, dy, Rsphere=1.0):
        q = np.zeros((4, mx + 2 * mbc, my + 2 * mbc), order="F")
        aux = np.zeros((16, mx + 2 * mbc, my + 2 * mbc), order="F")
        ci = lambda v: C.byref(C.c_int(v))
        cd = lambda v: C.byref(C.c_double(v))
        self.lib.qinit_(ci(mx), ci(my), ci(4), ci(mbc), ci(mx), ci(my), cd(xlower), cd(ylower), cd(dx), cd(dy), _d(q),
                        ci(16), _d(aux), cd(Rsphere))
        return q

    def sphere_src2(self, q, aux, xlower, ylower, dx, dy, dt, Rsphere=1.0):
        assert q.flags.f_contiguous and aux.flags.f_contiguous
        meqn, mx, my = q.shape
        ci = lambda v: C.byref(C.c_int(v))
        cd = lambda v: C.byref(C.c_double(v))
        self.lib.src2_(ci(mx), ci(my), ci(meqn), ci(2), ci(mx), ci(my), cd(xlower), cd(ylower), cd(dx), cd(dy), _d(q),
                       ci(aux.shape[0]), _d(aux), cd(0.0), cd(dt), cd(Rsphere))
        return q

    def qcor(self, ixy, i, aux1d, q1d, mbc, g, dx, dy):
        """qcor(ixy,i,m,aux,q,maxm,meqn,mbc,qc): aux1d (16, maxm+2mbc), q1d (meqn, maxm+2mbc); i is the Fortran index"""
        class _Comxyt(C.Structure):
            _fields_ = [("dtcom", C.c_double), ("dxcom", C.c_double), ("dycom", C.c_double), ("tcom", C.c_double),
                        ("icom", C.c_int), ("jcom", C.c_int)]
        _Comxyt.in_dll(self.lib, "comxyt_").dxcom = dx
        _Comxyt.in_dll(self.lib, "comxyt_").dycom = dy
        C.c_double.in_dll(self.lib, "sw_").value = g
        meqn, n = q1d.shape
        qc = np.zeros(4)
        ci = lambda v: C.byref(C.c_int(v))
        self.lib.qcor_(ci(ixy), ci(i), ci(0), _d(_f64(aux1d)), _d(_f64(q1d)), ci(n - 2 * mbc), ci(meqn), ci(mbc), _d(qc))
        return qc
