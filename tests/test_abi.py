"""CPU-side checks of the drop-in boundary: libchebhip.so builds for gfx950, loads, and exports
every symbol include/chebhip.h declares; argument errors follow chebyshev.c:98,106,122; and
without a GPU the create calls fail loudly (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

import __graft_entry__ as ge

sp = ge.load()


@pytest.fixture(scope="module")
def L():
    ge.build()
    return sp.lib()


def _declared_functions():
    txt = open(os.path.join(sp.INCLUDE_DIR, "chebhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef[^;]*\(\s*\*[^;]*;", "", txt)          # function-pointer typedefs are not entry points
    return sorted(set(re.findall(r"\b([a-z_0-9]+)\s*\([^;{]*\)\s*;", txt)))


def test_header_symbols_exported(L):
    names = _declared_functions()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libchebhip.so does not export %s" % n
    assert sorted(sp.ABI_SYMBOLS) == names


def test_version_arch(L):
    assert L.chebhip_version() >= 100
    assert L.chebhip_arch() == b"gfx950"


def test_code_object_is_gfx950():
    blob = open(sp.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"cheb_sweep_kernel" in blob


def test_argument_errors(L):
    """chebyshev.c:98 (n<2), :106 (tr range), bad dims -> distinct nonzero codes, checked before any device use."""
    h = C.c_void_p()
    ints = lambda v: (C.c_int * len(v))(*v)
    assert L.cheb_plan_create(2, 2, ints([4, 4]), C.byref(h)) == 2
    assert b"tdim out of range" in L.chebhip_last_error()
    assert L.cheb_plan_create(2, -1, ints([4, 4]), C.byref(h)) == 2
    assert L.cheb_plan_create(1, 0, ints([1]), C.byref(h)) == 1
    assert b"must be >= 2" in L.chebhip_last_error()
    assert L.cheb_plan_create(2, 0, ints([4, 0]), C.byref(h)) == 3
    assert L.ell_op_create(0, ints([4]), C.byref(h)) == 3
    assert L.ell_op_create(11, ints([4] * 11), C.byref(h)) == 3
    assert L.cheb_apply(None, None, None, None) == 4
    assert L.ell_op_mult(None, None, None, None) == 4
    assert h.value is None


def test_no_cpu_fallback(L):
    """On a box without a GPU the product refuses to run instead of silently computing on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = L.cheb_plan_create(1, 0, (C.c_int * 1)(8), C.byref(h))
    assert rc == 5 and h.value is None
    assert b"no CPU fallback" in L.chebhip_last_error()
    with pytest.raises(sp.ChebhipError):
        sp.EllipticOp((8, 8))


def test_argument_errors_of_the_wider_abi(L):
    """Slab-mode creates, the Krylov driver and the slab copies check their arguments before touching a device."""
    h = C.c_void_p()
    ints = lambda v: (C.c_int * len(v))(*v)
    cb = sp.DIM0_FN(lambda *a: 0)
    cbp = C.cast(cb, C.c_void_p)
    assert L.chebhip_fgmres_create(-1, 30, C.byref(h)) == 1                     # n < 0 (0 = a rank without unknowns)
    assert L.chebhip_fgmres_create(100, 0, C.byref(h)) == 4                     # restart out of range
    assert L.chebhip_fgmres_create(100, 100000, C.byref(h)) == 4
    assert L.chebhip_fgmres_set_tolerances(None, 1e-5, 1e-50, 10) == 4
    assert L.chebhip_fgmres_solve(None, None, None, None, None, None, None, 0, None) == 4
    assert L.ell_op_create_slab(1, ints([8]), 0, 4, cbp, None, C.byref(h)) == 3              # slab mode needs d >= 2
    assert L.ell_op_create_slab(2, ints([8, 8]), 5, 3, cbp, None, C.byref(h)) == 4           # empty / reversed plane range
    assert L.ell_op_create_slab(2, ints([8, 8]), 0, 9, cbp, None, C.byref(h)) == 4           # beyond the grid
    assert L.ell_op_create_slab(2, ints([8, 8]), 0, 4, None, None, C.byref(h)) == 4          # no callback
    assert L.stokes_op_create_slab(3, ints([8, 8, 8]), 4, 4, cbp, None, C.byref(h)) == 4
    assert L.stokes_op_create_slab(4, ints([8, 8, 8, 8]), 0, 4, cbp, None, C.byref(h)) == 3  # d = 2 or 3 (stokes.C:1036)
    assert L.stokes_op_create_slab(3, ints([8, 8, 8]), 0, 4, None, None, C.byref(h)) == 4
    assert L.stokes_op_mult_schur(None, None, None, None, None, None) == 4
    longs = lambda v: (C.c_long * len(v))(*v)
    one = C.c_void_p(8)                                                          # never dereferenced: the checks come first
    assert L.cheb_slab_pack(4, 6, 2, 2, longs([0, 3, 5]), one, one, None) == 4   # splits do not cover 0..M1
    assert L.cheb_slab_pack(4, 6, 2, 2, longs([0, 7, 6]), one, one, None) == 4   # not monotone / out of range
    assert L.cheb_slab_pack(4, 6, 2, 0, longs([0]), one, one, None) == 4         # G < 1
    assert L.cheb_slab_unpack_add(4, 6, 2, 2, longs([0, 3, 6]), None, None, 1.0, one, None) == 4
    # the process-rank (IPC) transport: names, ranks and groups are checked before any shared memory or device is touched
    assert L.chebhip_ipc_group_open(None, 2, 0, C.byref(h)) == 4
    assert L.chebhip_ipc_group_open(b"no-leading-slash", 2, 0, C.byref(h)) == 4
    assert L.chebhip_ipc_group_open(b"/chebhip-test-abi", 2, 2, C.byref(h)) == 4          # rank out of range
    assert L.chebhip_ipc_group_open(b"/chebhip-test-abi", 65, 0, C.byref(h)) == 4
    assert L.chebhip_comm_create_ipc(None, None, C.byref(h)) == 4
    assert L.chebhip_ipc_group_close(None) == 0 and L.chebhip_ipc_group_abort(None) == 0
    assert L.chebhip_comm_null_set_shadow(None, 0, None) == 4                               # not a NULL-transport communicator
    c = C.c_void_p()
    assert L.chebhip_comm_create_null(4, 1, C.byref(c)) == 0 and c.value
    assert L.chebhip_comm_null_set_shadow(c, 2, None) == 4 and L.chebhip_comm_null_set_shadow(c, 0, None) == 0
    assert L.chebhip_comm_destroy(c) == 0
    assert h.value is None
