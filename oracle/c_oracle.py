"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): ctypes binding of
oracle/liboracle_c.so, the plain-C restatement of the batched HMC transition."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle_c.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'oracle_c.c')
    if force or not os.path.exists(_SO) or \
            os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'liboracle_c.so'],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_pairwise_sum.restype = ctypes.c_double
        _lib.oracle_pairwise_sum.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        _lib.oracle_np_sum.restype = ctypes.c_double
        _lib.oracle_np_sum.argtypes = [ctypes.c_void_p, ctypes.c_int64]
        for fn in (_lib.oracle_hmc_sample_gauss, _lib.oracle_hmc_sample_gauss_fma):
            fn.restype = ctypes.c_int
            fn.argtypes = (
                [ctypes.c_void_p] * 8 + [ctypes.c_int64, ctypes.c_int64,
                                         ctypes.c_int32, ctypes.c_double,
                                         ctypes.c_double, ctypes.c_int32,
                                         ctypes.c_double, ctypes.c_double,
                                         ctypes.c_int32])
        _lib.oracle_polyval.restype = ctypes.c_int
        _lib.oracle_polyval.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int64] * 3
        _lib.oracle_poly_gauss_logp.restype = ctypes.c_int
        _lib.oracle_poly_gauss_logp.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64] * 3
    return _lib


def polyval(xs, coeffs):
    """``[C, N]``: numpy.polynomial.polynomial.polyval(xs, coeffs[c]) per row (oracle_c.c)."""
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    co = np.ascontiguousarray(np.atleast_2d(coeffs), dtype=np.float64)
    out = np.empty((co.shape[0], xs.size))
    rc = lib().oracle_polyval(xs.ctypes.data, co.ctypes.data, out.ctypes.data, co.shape[0], co.shape[1], xs.size)
    if rc != 0:
        raise ValueError('oracle_polyval rc=%d' % rc)
    return out


def poly_gauss_logp(coeffs, xs, ys, precision):
    """(log_prob [C], chi2 [C]) of the polynomial + Gaussian likelihood (oracle_c.c)."""
    xs = np.ascontiguousarray(xs, dtype=np.float64)
    ys = np.ascontiguousarray(ys, dtype=np.float64)
    co = np.ascontiguousarray(np.atleast_2d(coeffs), dtype=np.float64)
    C = co.shape[0]
    pr = np.ascontiguousarray(np.broadcast_to(np.asarray(precision, dtype=np.float64), (C,))).copy()
    out, chi2 = np.empty(C), np.empty(C)
    rc = lib().oracle_poly_gauss_logp(co.ctypes.data, xs.ctypes.data, ys.ctypes.data, pr.ctypes.data,
                                      out.ctypes.data, chi2.ctypes.data, C, co.shape[1], xs.size)
    if rc != 0:
        raise ValueError('oracle_poly_gauss_logp rc=%d' % rc)
    return out, chi2


def np_sum(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().oracle_np_sum(a.ctypes.data, a.size)


def hmc_sample_gauss(q0, p0, u, timestep, nsteps, k=1.0, x0=0.0, adapt=False,
                     uprate=1.05, downrate=0.95, nthreads=1, fma=False):
    """``fma=True``: the package's FMA mode (each leapfrog update one C99 ``fma``)."""
    q0 = np.ascontiguousarray(q0, dtype=np.float64)
    p0 = np.ascontiguousarray(p0, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    C, D = q0.shape
    assert p0.shape == (C, D) and u.shape == (C,)
    dt = np.ascontiguousarray(
        np.broadcast_to(np.asarray(timestep, dtype=np.float64), (C,))).copy()
    q_out = np.empty_like(q0)
    acc = np.zeros(C, dtype=np.uint8)
    eb = np.empty(C)
    ea = np.empty(C)
    fn = lib().oracle_hmc_sample_gauss_fma if fma else lib().oracle_hmc_sample_gauss
    rc = fn(
        q0.ctypes.data, p0.ctypes.data, u.ctypes.data, q_out.ctypes.data,
        acc.ctypes.data, eb.ctypes.data, ea.ctypes.data, dt.ctypes.data,
        C, D, int(nsteps), float(k), float(x0), int(bool(adapt)),
        float(uprate), float(downrate), int(nthreads))
    if rc != 0:
        raise ValueError('oracle_hmc_sample_gauss rc=%d' % rc)
    return dict(q_out=q_out, accepted=acc, e_before=eb, e_after=ea,
                timestep_out=dt)
