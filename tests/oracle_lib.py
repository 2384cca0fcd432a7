"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product path.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(_ORACLE_DIR, "liboracle.so")

DIRECT, FAST = 0, 1


def build(force=False):
    src = os.path.join(_ORACLE_DIR, "cheb_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        ip = C.POINTER(C.c_int)
        _lib.orc_redft00.argtypes = [C.c_int, dp, C.c_long, dp, C.c_long, C.c_int]
        _lib.orc_rodft00.argtypes = [C.c_int, dp, C.c_long, dp, C.c_long, C.c_int]
        _lib.orc_cheb_mult.argtypes = [C.c_int, C.c_int, ip, dp, dp, C.c_int, C.c_int]
        _lib.orc_cheb_mult_truth.argtypes = [C.c_int, C.c_int, ip, dp, dp]
        for f in (_lib.orc_local_size, _lib.orc_global_size, _lib.orc_dirichlet_size):
            f.argtypes = [C.c_int, ip]
            f.restype = C.c_long
        _lib.orc_elliptic_mult.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp, C.c_int, C.c_int]
        _lib.orc_elliptic_function.argtypes = [C.c_int, ip, C.c_double, C.c_double, dp, dp, dp,
                                               dp, dp, dp, dp, C.c_int, C.c_int]
        _lib.orc_elliptic_exact.argtypes = [C.c_int, ip, C.c_int, C.c_double, C.c_double, C.c_double,
                                            dp, dp, dp]
    return _lib


def _dp(a):
    if a is None:
        return None
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(dims):
    return (C.c_int * len(dims))(*[int(v) for v in dims])


def redft00(x, mode=DIRECT):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    assert lib().orc_redft00(x.size, _dp(x), 1, _dp(y), 1, mode) == 0
    return y


def rodft00(x, mode=DIRECT):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    assert lib().orc_rodft00(x.size, _dp(x), 1, _dp(y), 1, mode) == 0
    return y


def cheb_mult(x, tr, mode=FAST, nthreads=1):
    """ChebMult (chebyshev.c:142-199) along axis `tr` of the C-ordered array x."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    err = lib().orc_cheb_mult(x.ndim, tr, _ip(x.shape), _dp(x), _dp(y), mode, nthreads)
    if err:
        raise ValueError("orc_cheb_mult error %d" % err)
    return y


def cheb_mult_truth(x, tr):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    err = lib().orc_cheb_mult_truth(x.ndim, tr, _ip(x.shape), _dp(x), _dp(y))
    if err:
        raise ValueError("orc_cheb_mult_truth error %d" % err)
    return y


def sizes(dims):
    d = len(dims)
    return (lib().orc_local_size(d, _ip(dims)), lib().orc_global_size(d, _ip(dims)),
            lib().orc_dirichlet_size(d, _ip(dims)))


def elliptic_mult(dims, U, eta=None, deta=None, gradu0=None, mode=FAST, nthreads=1):
    """MatMult_Elliptic (elliptic.C:297-339)."""
    d = len(dims)
    N, G, _ = sizes(dims)
    eta = np.ones(N) if eta is None else np.ascontiguousarray(eta, dtype=np.float64)
    deta = np.zeros(N) if deta is None else np.ascontiguousarray(deta, dtype=np.float64)
    gradu0 = np.zeros(d * N) if gradu0 is None else np.ascontiguousarray(gradu0, dtype=np.float64).ravel()
    U = np.ascontiguousarray(U, dtype=np.float64)
    assert U.size == G
    V = np.empty(G)
    err = lib().orc_elliptic_mult(d, _ip(dims), _dp(eta), _dp(deta), _dp(gradu0), _dp(U), _dp(V), mode, nthreads)
    if err:
        raise ValueError("orc_elliptic_mult error %d" % err)
    return V


def elliptic_function(dims, U, b=None, dirichlet=None, gamma=0.0, exponent=2.0, mode=FAST, nthreads=1):
    """FormFunction (elliptic.C:481-533); returns rhs, eta, deta, gradu."""
    d = len(dims)
    N, G, _ = sizes(dims)
    U = np.ascontiguousarray(U, dtype=np.float64)
    rhs = np.empty(G)
    eta = np.empty(N)
    deta = np.empty(N)
    gradu = np.empty(d * N)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    dirichlet = None if dirichlet is None else np.ascontiguousarray(dirichlet, dtype=np.float64)
    err = lib().orc_elliptic_function(d, _ip(dims), gamma, exponent, _dp(dirichlet), _dp(U), _dp(b),
                                      _dp(rhs), _dp(eta), _dp(deta), _dp(gradu), mode, nthreads)
    if err:
        raise ValueError("orc_elliptic_function error %d" % err)
    return rhs, eta, deta, gradu.reshape(d, N)


def elliptic_exact(dims, exact, gamma=0.0, exponent=2.0, cos_scale=1.0):
    """CreateExactSolution (elliptic.C:594-677): u, u2 (global), dirichlet (compact)."""
    d = len(dims)
    _, G, D = sizes(dims)
    u = np.empty(G)
    u2 = np.empty(G)
    dv = np.empty(D)
    err = lib().orc_elliptic_exact(d, _ip(dims), exact, gamma, exponent, cos_scale, _dp(u), _dp(u2), _dp(dv))
    if err:
        raise ValueError("orc_elliptic_exact error %d" % err)
    return u, u2, dv
