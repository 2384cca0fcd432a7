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

DIRECT, FAST, FFTW = 0, 1, 2      # FFTW: genuine libfftw3 through dlopen, where the box has it (fftw_available())


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


def fftw_available():
    return bool(lib().orc_fftw_available())


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


def elliptic_mult_timed(dims, U, mode=FAST, nthreads=1, warm=2, reps=5):
    """The linear MatMult_Elliptic applied warm + reps times (bench.py cpu_baseline, BASELINE.md section 3 protocol);
    returns (V, [seconds of each timed apply])."""
    d = len(dims)
    N, G, _ = sizes(dims)
    U = np.ascontiguousarray(U, dtype=np.float64)
    V = np.empty(G)
    secs = np.zeros(reps)
    f = lib().orc_elliptic_mult_timed
    f.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    err = f(d, _ip(dims), _dp(U), _dp(V), mode, nthreads, warm, reps, _dp(secs))
    if err:
        raise ValueError("orc_elliptic_mult_timed error %d" % err)
    return V, [float(v) for v in secs]


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


def fd_matrix(dims, eta=None, deta=None, gradu=None):
    """FormJacobian's matrix P (elliptic.C:537-590) as a scipy CSR matrix on the global (interior) vector; gradu None
    and deta ignored: one velocity component of MatVVPC (stokes.C:1181-1226)."""
    import scipy.sparse as sps
    d = len(dims)
    N, G, _ = sizes(dims)
    eta = np.ones(N) if eta is None else np.ascontiguousarray(eta, dtype=np.float64).ravel()
    deta = np.zeros(N) if deta is None else np.ascontiguousarray(deta, dtype=np.float64).ravel()
    gu = None if gradu is None else np.ascontiguousarray(gradu, dtype=np.float64).ravel()
    W = 2 * d + 1
    cols = np.empty(G * W, dtype=np.int32); vals = np.empty(G * W)
    L = lib()
    L.orc_fd_matrix.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_int), C.POINTER(C.c_double)]
    err = L.orc_fd_matrix(d, _ip(dims), _dp(eta), _dp(deta), _dp(gu), cols.ctypes.data_as(C.POINTER(C.c_int)), _dp(vals))
    if err:
        raise ValueError("orc_fd_matrix error %d" % err)
    rows = np.repeat(np.arange(G), W)
    keep = cols >= 0                                            # MatSetValues ignores negative indices
    return sps.csr_matrix((vals[keep], (rows[keep], cols[keep])), shape=(G, G))


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


# ---------------------------------------------------------------------------------------------
# Stokes (stokes.C), -boundary 0
# ---------------------------------------------------------------------------------------------
class Rheology(C.Structure):
    _fields_ = [("kind", C.c_int), ("hardness", C.c_double), ("exponent", C.c_double),
                ("regularization", C.c_double), ("gamma0", C.c_double)]


def _bind_stokes():
    L = lib()
    if getattr(L, "_stokes_bound", False):
        return L
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    L.orc_stokes_pressure_reduce.argtypes = [C.c_int, ip, dp]
    L.orc_stokes_mult_vv.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp, C.c_int, C.c_int]
    L.orc_stokes_divergence.argtypes = [C.c_int, ip, dp, dp, dp, C.c_int, C.c_int]
    L.orc_stokes_mult_vp.argtypes = [C.c_int, ip, dp, dp, C.c_int, C.c_int]
    L.orc_stokes_mult.argtypes = [C.c_int, ip, dp, dp, dp, dp, dp, C.c_int, C.c_int]
    L.orc_stokes_function.argtypes = [C.c_int, ip, C.POINTER(Rheology), dp, dp, dp, dp, dp, dp, dp, C.c_int, C.c_int]
    L.orc_stokes_exact.argtypes = [C.c_int, ip, C.c_int, dp, dp, dp]
    L._stokes_bound = True
    return L


def stokes_sizes(dims):
    """(N local nodes, I interior nodes, gv, gp, g, dv)."""
    d = len(dims)
    N, I, D = sizes(dims)
    return N, I, d * I, I, (d + 1) * I, d * D


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64)


def _state(dims, eta, deta, strain):
    d = len(dims)
    N = sizes(dims)[0]
    eta = np.ones(N) if eta is None else _c(eta).ravel()
    deta = np.zeros(N) if deta is None else _c(deta).ravel()
    strain = np.zeros(d * N * d) if strain is None else _c(strain).ravel()
    assert eta.size == N and deta.size == N and strain.size == d * N * d
    return eta, deta, strain


def stokes_pressure_reduce(dims, pL):
    pL = _c(pL).copy().ravel()
    err = _bind_stokes().orc_stokes_pressure_reduce(len(dims), _ip(dims), _dp(pL))
    if err:
        raise ValueError("orc_stokes_pressure_reduce error %d" % err)
    return pL


def stokes_mult_vv(dims, vG, eta=None, deta=None, strain=None, mode=FAST, nthreads=1):
    eta, deta, strain = _state(dims, eta, deta, strain)
    vG = _c(vG)
    out = np.empty_like(vG)
    err = _bind_stokes().orc_stokes_mult_vv(len(dims), _ip(dims), _dp(eta), _dp(deta), _dp(strain), _dp(vG), _dp(out), mode, nthreads)
    if err:
        raise ValueError("orc_stokes_mult_vv error %d" % err)
    return out


def stokes_divergence(dims, vG, dirichlet=None, mode=FAST, nthreads=1):
    vG = _c(vG)
    out = np.empty(stokes_sizes(dims)[3])
    err = _bind_stokes().orc_stokes_divergence(len(dims), _ip(dims), _dp(_c(dirichlet)), _dp(vG), _dp(out), mode, nthreads)
    if err:
        raise ValueError("orc_stokes_divergence error %d" % err)
    return out


def stokes_mult_vp(dims, pG, mode=FAST, nthreads=1):
    pG = _c(pG)
    out = np.empty(stokes_sizes(dims)[2])
    err = _bind_stokes().orc_stokes_mult_vp(len(dims), _ip(dims), _dp(pG), _dp(out), mode, nthreads)
    if err:
        raise ValueError("orc_stokes_mult_vp error %d" % err)
    return out


def stokes_mult(dims, xG, eta=None, deta=None, strain=None, mode=FAST, nthreads=1):
    eta, deta, strain = _state(dims, eta, deta, strain)
    xG = _c(xG)
    out = np.empty_like(xG)
    err = _bind_stokes().orc_stokes_mult(len(dims), _ip(dims), _dp(eta), _dp(deta), _dp(strain), _dp(xG), _dp(out), mode, nthreads)
    if err:
        raise ValueError("orc_stokes_mult error %d" % err)
    return out


def stokes_function(dims, xG, dirichlet, force, rheology=(0, 1.0, 1.0, 1.0, 1.0), mode=FAST, nthreads=1):
    """StokesFunction (stokes.C:680-758); returns y, eta, deta, strain[d][N*d]."""
    d = len(dims)
    N = sizes(dims)[0]
    xG = _c(xG)
    y = np.empty_like(xG)
    eta, deta, strain = np.empty(N), np.empty(N), np.empty(d * N * d)
    rh = Rheology(*rheology)
    err = _bind_stokes().orc_stokes_function(d, _ip(dims), C.byref(rh), _dp(_c(dirichlet)), _dp(_c(force)), _dp(xG), _dp(y),
                                             _dp(eta), _dp(deta), _dp(strain), mode, nthreads)
    if err:
        raise ValueError("orc_stokes_function error %d" % err)
    return y, eta, deta, strain.reshape(d, N * d)


def stokes_exact(dims, exact):
    d = len(dims)
    _, _, _, _, g, dv = stokes_sizes(dims)
    U, U2, dvals = np.empty(g), np.empty(g), np.empty(dv)
    err = _bind_stokes().orc_stokes_exact(d, _ip(dims), exact, _dp(U), _dp(U2), _dp(dvals))
    if err:
        raise ValueError("orc_stokes_exact error %d" % err)
    return U, U2, dvals
