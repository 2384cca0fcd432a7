"""Host-side mirror of the reference interface over the C ABI of libchebhip.so.

The product is the C-ABI shared library (include/chebhip.h); this module is the
thin ctypes binding the tests and bench.py use, shaped after the reference's
operator interface:

  ChebPlan(dims, tr).mult(x, y)        <-> MatCreateCheb / ChebMult / ChebDestroy
                                           (chebyshev.h:31-34, chebyshev.c:89-235)
  EllipticOp(dims).mult(U, V)          <-> MatCreate_Elliptic / MatMult_Elliptic
                                           (elliptic.C:250-339)
  EllipticOp.function(U, b, rhs, ...)  <-> FormFunction (elliptic.C:481-533)

Device vectors are torch.float64 CUDA(HIP) tensors; torch is used only for
device memory and streams.  There is no CPU fallback: if libchebhip.so is
missing or no GPU is usable, construction raises.

The directory name carries a hyphen, so import it with `load()` from
__graft_entry__.py (importlib by path) rather than `import`.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHEBHIP_LIB_PATH: a diagnostic build of the same library (tools/v4_ablate.sh, tools/f4_ablate.sh); production uses the in-tree one
LIB_PATH = os.environ.get("CHEBHIP_LIB_PATH") or os.path.join(_HERE, "libchebhip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# every symbol include/chebhip.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "chebhip_last_error", "chebhip_version", "chebhip_arch", "chebhip_launch_count",
    "chebhip_set_option", "chebhip_get_option", "chebhip_option_name",
    "cheb_plan_create", "cheb_apply", "cheb_apply_host", "cheb_plan_destroy", "cheb_plan_size",
    "cheb_plan_create_trimmed", "cheb_apply_lap1d", "cheb_slab_pack", "cheb_slab_unpack_add",
    "ell_op_create", "ell_op_destroy", "ell_op_local_size", "ell_op_global_size",
    "ell_op_dirichlet_size", "ell_op_mult", "ell_op_mult_host", "ell_op_function",
    "ell_op_function_host", "ell_op_set_dirichlet", "ell_op_get_state", "ell_op_set_state",
    "ell_op_create_slab", "ell_op_pencil_sweep",
    "stokes_op_create", "stokes_op_destroy", "stokes_op_size", "stokes_op_set_rheology",
    "stokes_op_set_dirichlet", "stokes_op_set_force", "stokes_op_mult", "stokes_op_mult_vv",
    "stokes_op_mult_pv", "stokes_op_mult_vp", "stokes_op_mult_vv_cm", "stokes_op_mult_pv_cm", "stokes_op_mult_vp_cm", "stokes_op_mult_schur_cm", "stokes_op_function", "stokes_op_get_state",
    "stokes_op_set_state", "stokes_op_create_slab", "stokes_op_pencil_sweep", "stokes_op_pencil_pressure", "stokes_op_pencil_sweep_pressure", "stokes_op_mult_schur", "stokes_op_set_inner_solver", "stokes_op_inner_iterations", "stokes_op_set_inner_reduce",
    "chebhip_fgmres_create", "chebhip_fgmres_destroy", "chebhip_fgmres_set_tolerances", "chebhip_fgmres_solve",
    "chebhip_fgmres_iterations", "chebhip_fgmres_residual", "chebhip_fgmres_reason", "chebhip_fgmres_set_reduce",
    "ell_pc_create", "stokes_pc_create", "chebhip_fdpc_destroy", "chebhip_fdpc_update", "chebhip_fdpc_set_sweeps",
    "chebhip_fdpc_mult", "chebhip_fdpc_apply", "chebhip_fdpc_apply_cm",
    "stokes_saddle_create", "stokes_saddle_destroy", "stokes_saddle_set_type", "stokes_saddle_set_inner",
    "stokes_saddle_setup", "stokes_saddle_apply", "stokes_saddle_iterations", "stokes_saddle_set_pc_sweeps", "stokes_saddle_set_schur_jacobi",
    "chebhip_timers_enable", "chebhip_timers_reset", "chebhip_timers_read", "chebhip_stage_name",
    "stokes_op_viscosity_range", "stokes_op_write_vtk",
    "chebhip_dist_create", "chebhip_dist_destroy", "chebhip_dist_local_size", "chebhip_dist_slab_offset",
    "chebhip_dist_use_rccl", "chebhip_dist_set_exchange", "chebhip_dist_mult", "chebhip_dist_mult_batch",
    "chebhip_rccl_unique_id", "chebhip_rccl_comm_create", "chebhip_rccl_comm_destroy", "chebhip_rccl_reduce",
    "chebhip_comm_create_rccl", "chebhip_local_group_create", "chebhip_local_group_destroy", "chebhip_local_group_abort",
    "chebhip_comm_create_local", "chebhip_comm_create_callback", "chebhip_comm_create_null", "chebhip_comm_null_set_shadow", "chebhip_comm_destroy", "chebhip_comm_size", "chebhip_comm_rank",
    "chebhip_ipc_group_open", "chebhip_ipc_group_close", "chebhip_ipc_group_abort", "chebhip_comm_create_ipc",
    "chebhip_comm_reduce", "chebhip_dist_use_comm",
    "chebhip_dist_stokes_create", "chebhip_dist_stokes_destroy", "chebhip_dist_stokes_op", "chebhip_dist_stokes_ranges",
    "chebhip_dist_ell_create", "chebhip_dist_ell_destroy", "chebhip_dist_ell_op", "chebhip_dist_ell_ranges",
    "stokes_pc_create_slab", "ell_pc_create_slab", "chebhip_fdpc_pencil_transform", "chebhip_dist_stokes_pc", "chebhip_dist_ell_pc",
    "stokes_saddle_create_slab",
]


class ChebhipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("chebhip error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile libchebhip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "-s", "clean"])
    subprocess.check_call(["make", "-C", src, "-s"])
    return LIB_PATH


_lib = None


def lib():
    """Load the C-ABI library; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ChebhipError(-1, "libchebhip.so not built (run __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
        L.chebhip_last_error.restype = C.c_char_p
        L.chebhip_arch.restype = C.c_char_p
        L.chebhip_launch_count.restype = C.c_long
        L.chebhip_set_option.argtypes = [C.c_char_p, C.c_int]
        L.chebhip_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int)]
        L.chebhip_option_name.argtypes = [C.c_int]
        L.chebhip_option_name.restype = C.c_char_p
        L.cheb_plan_create.argtypes = [C.c_int, C.c_int, ip, C.POINTER(vp)]
        L.cheb_apply.argtypes = [vp, vp, vp, vp]
        L.cheb_plan_create_trimmed.argtypes = [C.c_int, C.c_int, ip, C.POINTER(vp)]
        L.cheb_apply_lap1d.argtypes = [vp, vp, vp, C.c_double, vp, vp]
        lp = C.POINTER(C.c_long)
        L.cheb_slab_pack.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, lp, vp, vp, vp]
        L.cheb_slab_unpack_add.argtypes = [C.c_long, C.c_long, C.c_long, C.c_int, lp, vp, vp, C.c_double, vp, vp]
        L.cheb_apply_host.argtypes = [vp, dp, dp]
        L.cheb_plan_destroy.argtypes = [vp]
        L.cheb_plan_size.argtypes = [vp]
        L.cheb_plan_size.restype = C.c_long
        L.ell_op_create.argtypes = [C.c_int, ip, C.POINTER(vp)]
        L.ell_op_destroy.argtypes = [vp]
        L.ell_op_create_slab.argtypes = [C.c_int, ip, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
        L.ell_op_pencil_sweep.argtypes = [vp, C.c_long, vp, vp, vp]
        for f in (L.ell_op_local_size, L.ell_op_global_size, L.ell_op_dirichlet_size):
            f.argtypes = [vp]
            f.restype = C.c_long
        L.ell_op_mult.argtypes = [vp, vp, vp, vp]
        L.ell_op_mult_host.argtypes = [vp, dp, dp]
        L.ell_op_function.argtypes = [vp, C.c_double, C.c_double, vp, vp, vp, vp]
        L.ell_op_function_host.argtypes = [vp, C.c_double, C.c_double, dp, dp, dp]
        L.ell_op_set_dirichlet.argtypes = [vp, dp]
        L.ell_op_get_state.argtypes = [vp, C.c_int, dp]
        L.ell_op_set_state.argtypes = [vp, C.c_int, dp]
        L.stokes_op_create.argtypes = [C.c_int, ip, C.POINTER(vp)]
        L.stokes_op_destroy.argtypes = [vp]
        L.stokes_op_size.argtypes = [vp, C.c_int]
        L.stokes_op_size.restype = C.c_long
        L.stokes_op_set_rheology.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.stokes_op_set_dirichlet.argtypes = [vp, dp]
        L.stokes_op_set_force.argtypes = [vp, dp]
        for f in (L.stokes_op_mult, L.stokes_op_mult_vv, L.stokes_op_mult_pv, L.stokes_op_mult_vp, L.stokes_op_function,
                  L.stokes_op_mult_vv_cm, L.stokes_op_mult_pv_cm, L.stokes_op_mult_vp_cm):
            f.argtypes = [vp, vp, vp, vp]
        L.stokes_op_mult_schur_cm.argtypes = [vp, vp, vp, vp, vp, vp]
        L.stokes_op_get_state.argtypes = [vp, C.c_int, dp]
        L.stokes_op_set_state.argtypes = [vp, C.c_int, dp]
        L.stokes_op_create_slab.argtypes = [C.c_int, ip, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
        L.stokes_op_pencil_sweep.argtypes = [vp, C.c_int, C.c_long, vp, vp, vp]
        L.stokes_op_pencil_pressure.argtypes = [vp, C.c_long, vp, vp, vp]
        L.stokes_op_mult_schur.argtypes = [vp, vp, vp, vp, vp, vp]
        L.stokes_op_set_inner_solver.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_int]
        L.stokes_op_inner_iterations.argtypes = [vp]
        L.stokes_op_set_inner_reduce.argtypes = [vp, vp, vp]
        L.chebhip_fgmres_create.argtypes = [C.c_long, C.c_int, C.POINTER(vp)]
        L.chebhip_fgmres_destroy.argtypes = [vp]
        L.chebhip_fgmres_set_tolerances.argtypes = [vp, C.c_double, C.c_double, C.c_int]
        L.chebhip_fgmres_solve.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp]
        L.chebhip_fgmres_iterations.argtypes = [vp]
        L.chebhip_fgmres_residual.argtypes = [vp]
        L.chebhip_fgmres_residual.restype = C.c_double
        L.chebhip_fgmres_reason.argtypes = [vp]
        L.chebhip_fgmres_set_reduce.argtypes = [vp, vp, vp]
        L.ell_pc_create.argtypes = [vp, C.POINTER(vp)]
        L.stokes_pc_create.argtypes = [vp, C.POINTER(vp)]
        L.chebhip_fdpc_destroy.argtypes = [vp]
        L.chebhip_fdpc_update.argtypes = [vp, vp]
        L.chebhip_fdpc_set_sweeps.argtypes = [vp, C.c_int]
        L.chebhip_fdpc_mult.argtypes = [vp, vp, vp, vp]
        L.chebhip_fdpc_apply.argtypes = [vp, vp, vp, vp]
        L.chebhip_fdpc_apply_cm.argtypes = [vp, vp, vp, vp]
        L.stokes_saddle_create.argtypes = [vp, C.POINTER(vp)]
        L.stokes_saddle_destroy.argtypes = [vp]
        L.stokes_saddle_set_type.argtypes = [vp, C.c_int]
        L.stokes_saddle_set_inner.argtypes = [vp, C.c_int, C.c_int, C.c_double]
        L.stokes_saddle_setup.argtypes = [vp, vp]
        L.stokes_saddle_apply.argtypes = [vp, vp, vp, vp]
        L.stokes_saddle_iterations.argtypes = [vp, C.c_int]
        L.stokes_saddle_set_pc_sweeps.argtypes = [vp, C.c_int]
        L.stokes_saddle_set_schur_jacobi.argtypes = [vp, C.c_int]
        L.chebhip_timers_enable.argtypes = [C.c_int]
        L.chebhip_timers_read.argtypes = [C.c_int, dp, C.POINTER(C.c_long)]
        L.chebhip_stage_name.argtypes = [C.c_int]
        L.chebhip_stage_name.restype = C.c_char_p
        L.stokes_op_viscosity_range.argtypes = [vp, dp, dp, vp]
        L.stokes_op_write_vtk.argtypes = [vp, vp, C.c_char_p]
        L.chebhip_dist_create.argtypes = [C.c_int, ip, C.c_int, C.c_int, C.POINTER(vp)]
        L.chebhip_dist_destroy.argtypes = [vp]
        for f in (L.chebhip_dist_local_size, L.chebhip_dist_slab_offset):
            f.argtypes = [vp]
            f.restype = C.c_long
        L.chebhip_dist_use_rccl.argtypes = [vp, vp]
        L.chebhip_dist_set_exchange.argtypes = [vp, vp, vp]
        L.chebhip_dist_mult.argtypes = [vp, vp, vp, vp]
        L.chebhip_dist_mult_batch.argtypes = [vp, C.c_int, vp, vp, vp]
        L.chebhip_rccl_unique_id.argtypes = [vp]
        L.chebhip_rccl_comm_create.argtypes = [C.c_int, C.c_int, vp, C.POINTER(vp)]
        L.chebhip_rccl_comm_destroy.argtypes = [vp]
        L.chebhip_rccl_reduce.argtypes = [vp, vp, C.c_int, vp]
        L.chebhip_comm_create_rccl.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
        L.chebhip_local_group_create.argtypes = [C.c_int, C.POINTER(vp)]
        L.chebhip_local_group_destroy.argtypes = [vp]
        L.chebhip_local_group_abort.argtypes = [vp]
        L.chebhip_comm_create_local.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.chebhip_comm_create_callback.argtypes = [C.c_int, C.c_int, vp, vp, vp, C.POINTER(vp)]
        L.chebhip_comm_destroy.argtypes = [vp]
        L.chebhip_comm_create_null.argtypes = [C.c_int, C.c_int, C.POINTER(vp)]
        L.chebhip_comm_null_set_shadow.argtypes = [vp, C.c_int, C.POINTER(vp)]
        L.chebhip_ipc_group_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
        L.chebhip_ipc_group_close.argtypes = [vp]
        L.chebhip_ipc_group_abort.argtypes = [vp]
        L.chebhip_comm_create_ipc.argtypes = [vp, vp, C.POINTER(vp)]
        L.chebhip_comm_size.argtypes = [vp]
        L.chebhip_comm_rank.argtypes = [vp]
        L.chebhip_comm_reduce.argtypes = [vp, vp, C.c_int, vp]
        L.chebhip_dist_use_comm.argtypes = [vp, vp]
        for nm in ("stokes", "ell"):
            getattr(L, "chebhip_dist_%s_create" % nm).argtypes = [C.c_int, ip, vp, C.POINTER(vp)]
            getattr(L, "chebhip_dist_%s_destroy" % nm).argtypes = [vp]
            getattr(L, "chebhip_dist_%s_op" % nm).argtypes = [vp]
            getattr(L, "chebhip_dist_%s_op" % nm).restype = vp
            getattr(L, "chebhip_dist_%s_ranges" % nm).argtypes = [vp, lp]
            getattr(L, "chebhip_dist_%s_pc" % nm).argtypes = [vp, C.POINTER(vp)]
            getattr(L, "%s_pc_create_slab" % nm).argtypes = [vp, C.c_long, vp, vp, C.POINTER(vp)]
        L.chebhip_fdpc_pencil_transform.argtypes = [vp, C.c_int, C.c_int, C.c_long, vp, vp, vp]
        L.stokes_saddle_create_slab.argtypes = [vp, vp, vp, vp, C.POINTER(vp)]
        _lib = L
    return _lib


def set_option(name, value):
    """chebhip_set_option: the library's run-time switches (include/chebhip.h lists them); the environment is never read."""
    _chk(lib().chebhip_set_option(name.encode(), int(value)))


def get_option(name):
    v = C.c_int()
    _chk(lib().chebhip_get_option(name.encode(), C.byref(v)))
    return v.value


def options():
    out, i = {}, 0
    while True:
        n = lib().chebhip_option_name(i).decode()
        if not n:
            return out
        out[n] = get_option(n)
        i += 1


def _chk(rc):
    if rc != 0:
        raise ChebhipError(rc, lib().chebhip_last_error().decode())


def _ints(v):
    return (C.c_int * len(v))(*[int(x) for x in v])


def _np_dp(a):
    import numpy as np
    assert isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _dev_ptr(t, n):
    import torch
    assert isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
    assert t.numel() == n, "expected %d elements, got %d" % (n, t.numel())
    return t.data_ptr()


def _stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def device_view(ptr, n):
    """A float64 tensor over n doubles of device memory owned by someone else (no copy)."""
    import torch

    if not ptr or int(n) == 0:          # an empty vector (a slab that owns only boundary planes): the library may hand over NULL
        return torch.empty(0, dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()))

    class _Arr:
        pass
    a = _Arr()
    a.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2, "strides": None}
    return torch.as_tensor(a, device=torch.device("cuda", torch.cuda.current_device()))


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)


def allreduce_trampoline(group=None):
    """A chebhip_reduce_fn that sums device doubles over the ranks of `group` with torch.distributed."""
    import torch.distributed as dist

    def red(ctx, ptr, count, stream):
        try:
            t = device_view(ptr, count)
            if dist.get_backend(group) == "gloo":          # rehearsal on one GPU: stage through the host
                h = t.cpu(); dist.all_reduce(h, group=group); t.copy_(h)
            else:
                dist.all_reduce(t, group=group)
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 5
    return REDUCE_FN(red)


class ChebPlan:
    """y = d/dx_tr x on a row-major tensor of shape dims (MatCreateCheb, chebyshev.c:89-138)."""

    def __init__(self, dims, tr):
        self.dims = tuple(int(d) for d in dims)
        self.tr = int(tr)
        h = C.c_void_p()
        _chk(lib().cheb_plan_create(len(self.dims), self.tr, _ints(self.dims), C.byref(h)))
        self._h = h
        self.size = lib().cheb_plan_size(h)

    def mult(self, x, y):
        """ChebMult (chebyshev.c:142-199) on device tensors, asynchronous on torch's current stream."""
        _chk(lib().cheb_apply(self._h, _dev_ptr(x, self.size), _dev_ptr(y, self.size), _stream()))
        return y

    def mult_host(self, x):
        import numpy as np
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.size
        y = np.empty_like(x)
        _chk(lib().cheb_apply_host(self._h, _np_dp(x), _np_dp(y)))
        return y

    def destroy(self):
        if self._h:
            lib().cheb_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Lap1dPlan:
    """y = acc + alpha * D_tr D_tr x on an interior-layout tensor (cheb_plan_create_trimmed /
    cheb_apply_lap1d): one direction of the linear MatMult_Elliptic, usable on slabs and pencils."""

    def __init__(self, dims, tr):
        self.dims = tuple(int(d) for d in dims)
        self.tr = int(tr)
        h = C.c_void_p()
        _chk(lib().cheb_plan_create_trimmed(len(self.dims), self.tr, _ints(self.dims), C.byref(h)))
        self._h = h
        self.size = lib().cheb_plan_size(h)

    def apply(self, x, y, acc=None, alpha=1.0):
        ap = _dev_ptr(acc, self.size) if acc is not None else None
        _chk(lib().cheb_apply_lap1d(self._h, _dev_ptr(x, self.size), ap, alpha, _dev_ptr(y, self.size), _stream()))
        return y

    def destroy(self):
        if self._h:
            lib().cheb_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def slab_pack(slab, buf, m0, M1, R, c1):
    """buf <- slab (m0, M1, R) reordered into per-peer column blocks (cheb_slab_pack)."""
    n = int(m0) * int(M1) * int(R)
    cs = (C.c_long * len(c1))(*[int(v) for v in c1])
    _chk(lib().cheb_slab_pack(m0, M1, R, len(c1) - 1, cs, _dev_ptr(slab, n), _dev_ptr(buf, n), _stream()))
    return buf


def slab_unpack_add(buf, acc, out, m0, M1, R, c1, alpha=1.0):
    """out = acc + alpha * slab-ordered(buf) (cheb_slab_unpack_add); acc may be None."""
    n = int(m0) * int(M1) * int(R)
    cs = (C.c_long * len(c1))(*[int(v) for v in c1])
    ap = _dev_ptr(acc, n) if acc is not None else None
    _chk(lib().cheb_slab_unpack_add(m0, M1, R, len(c1) - 1, cs, _dev_ptr(buf, n), ap, alpha, _dev_ptr(out, n), _stream()))
    return out


DIM0_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p)


def _dim0_trampoline(dim0):
    def tramp(ctx, kind, nf, inp, acc, alpha, out, stream):
        try:
            return int(dim0(kind, nf, inp, acc, alpha, out, stream) or 0)
        except Exception:                      # never unwind through the C frames
            import traceback
            traceback.print_exc()
            return 5
    return DIM0_FN(tramp)


class EllipticOp:
    """The scalar elliptic MatShell (MatCreate_Elliptic, elliptic.C:250-293)."""

    def __init__(self, dims, slab=None, dim0=None, handle=None):
        """slab = (lo, hi): the planes [lo, hi) of grid dimension 0 (ell_op_create_slab); dim0 is then the Python
        callable (kind, nfields, in_ptr, acc_ptr_or_None, alpha, out_ptr, stream) -> int doing the sweeps along dim 0.
        handle: wrap an ell_op owned by someone else (chebhip_dist_ell_op)."""
        self.dims = tuple(int(d) for d in dims)
        h = C.c_void_p()
        self._owned = handle is None
        if handle is not None:
            h = C.c_void_p(handle)
        elif slab is None:
            _chk(lib().ell_op_create(len(self.dims), _ints(self.dims), C.byref(h)))
        else:
            self._cb = _dim0_trampoline(dim0)           # keep the trampoline alive as long as the handle
            _chk(lib().ell_op_create_slab(len(self.dims), _ints(self.dims), int(slab[0]), int(slab[1]),
                                          C.cast(self._cb, C.c_void_p), None, C.byref(h)))
        self._h = h
        self.local_size = lib().ell_op_local_size(h)
        self.global_size = lib().ell_op_global_size(h)
        self.dirichlet_size = lib().ell_op_dirichlet_size(h)

    def mult(self, U, V):
        """MatMult_Elliptic (elliptic.C:297-339) on device tensors of global_size."""
        _chk(lib().ell_op_mult(self._h, _dev_ptr(U, self.global_size), _dev_ptr(V, self.global_size), _stream()))
        return V

    def pencil_sweep(self, ncol, inp, out):
        _chk(lib().ell_op_pencil_sweep(self._h, ncol, inp.data_ptr(), out.data_ptr(), _stream()))
        return out

    def mult_host(self, U):
        import numpy as np
        U = np.ascontiguousarray(U, dtype=np.float64)
        assert U.size == self.global_size
        V = np.empty_like(U)
        _chk(lib().ell_op_mult_host(self._h, _np_dp(U), _np_dp(V)))
        return V

    def function(self, U, b, rhs, gamma=0.0, exponent=2.0):
        """FormFunction (elliptic.C:481-533) on device tensors; b may be None."""
        bp = _dev_ptr(b, self.global_size) if b is not None else None
        _chk(lib().ell_op_function(self._h, gamma, exponent, _dev_ptr(U, self.global_size), bp,
                                   _dev_ptr(rhs, self.global_size), _stream()))
        return rhs

    def function_host(self, U, b=None, gamma=0.0, exponent=2.0):
        import numpy as np
        U = np.ascontiguousarray(U, dtype=np.float64)
        rhs = np.empty_like(U)
        bp = None
        if b is not None:
            b = np.ascontiguousarray(b, dtype=np.float64)
            bp = _np_dp(b)
        _chk(lib().ell_op_function_host(self._h, gamma, exponent, _np_dp(U), bp, _np_dp(rhs)))
        return rhs

    def set_dirichlet(self, values):
        import numpy as np
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert values.size == self.dirichlet_size
        _chk(lib().ell_op_set_dirichlet(self._h, _np_dp(values)))

    def get_state(self, which):
        import numpy as np
        out = np.empty(self.local_size)
        _chk(lib().ell_op_get_state(self._h, which, _np_dp(out)))
        return out

    def set_state(self, which, values):
        import numpy as np
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert values.size == self.local_size
        _chk(lib().ell_op_set_state(self._h, which, _np_dp(values)))

    def destroy(self):
        if self._h:
            if self._owned:
                lib().ell_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class StokesOp:
    """The Stokes MatShells (StokesCreate, stokes.C:257-344) with -boundary 0.

    mult <-> StokesMatMult (stokes.C:499-519); mult_vv / mult_pv / mult_vp <-> MatVV / MatPV / MatVP
    (:623-676, :557-566, :599-619); function <-> StokesFunction (:680-758)."""

    def __init__(self, dims, slab=None, dim0=None, handle=None):
        """slab = (lo, hi): the planes [lo, hi) of grid dimension 0 (stokes_op_create_slab); dim0 is then the Python
        callable (kind, nfields, in_ptr, acc_ptr_or_None, alpha, out_ptr, stream) -> int doing the work along dim 0.
        handle: wrap a stokes_op owned by someone else (chebhip_dist_stokes_op)."""
        self.dims = tuple(int(d) for d in dims)
        self.d = len(self.dims)
        h = C.c_void_p()
        self._owned = handle is None
        if handle is not None:
            h = C.c_void_p(handle)
        elif slab is None:
            _chk(lib().stokes_op_create(self.d, _ints(self.dims), C.byref(h)))
        else:
            self._cb = _dim0_trampoline(dim0)           # keep the trampoline alive as long as the handle
            _chk(lib().stokes_op_create_slab(self.d, _ints(self.dims), int(slab[0]), int(slab[1]),
                                             C.cast(self._cb, C.c_void_p), None, C.byref(h)))
        self._h = h
        sz = [lib().stokes_op_size(h, w) for w in range(6)]
        self.local_nodes, self.interior_nodes, self.velocity_size, self.pressure_size, self.global_size, self.dirichlet_size = sz

    def set_rheology(self, kind, hardness=1.0, exponent=1.0, regularization=1.0, gamma0=1.0):
        _chk(lib().stokes_op_set_rheology(self._h, kind, hardness, exponent, regularization, gamma0))

    def set_dirichlet(self, values):
        import numpy as np
        values = np.ascontiguousarray(values, dtype=np.float64)
        assert values.size == self.dirichlet_size
        _chk(lib().stokes_op_set_dirichlet(self._h, _np_dp(values)))

    def set_force(self, force):
        import numpy as np
        force = np.ascontiguousarray(force, dtype=np.float64)
        assert force.size == self.global_size
        _chk(lib().stokes_op_set_force(self._h, _np_dp(force)))

    def _call(self, fn, x, nx, y, ny):
        _chk(fn(self._h, _dev_ptr(x, nx), _dev_ptr(y, ny), _stream()))
        return y

    def mult(self, x, y):
        return self._call(lib().stokes_op_mult, x, self.global_size, y, self.global_size)

    def mult_vv(self, v, out):
        return self._call(lib().stokes_op_mult_vv, v, self.velocity_size, out, self.velocity_size)

    def mult_pv(self, v, pout):
        return self._call(lib().stokes_op_mult_pv, v, self.velocity_size, pout, self.pressure_size)

    def mult_vp(self, p, vout):
        return self._call(lib().stokes_op_mult_vp, p, self.pressure_size, vout, self.velocity_size)

    # the same on component-major velocity vectors (component c of interior node n at c * I + n): the layout of the block
    # preconditioners' inner solves
    def mult_vv_cm(self, v, out):
        return self._call(lib().stokes_op_mult_vv_cm, v, self.velocity_size, out, self.velocity_size)

    def mult_pv_cm(self, v, pout):
        return self._call(lib().stokes_op_mult_pv_cm, v, self.velocity_size, pout, self.pressure_size)

    def mult_vp_cm(self, p, vout):
        return self._call(lib().stokes_op_mult_vp_cm, p, self.pressure_size, vout, self.velocity_size)

    def pencil_sweep(self, nfields, ncol, inp, out):
        _chk(lib().stokes_op_pencil_sweep(self._h, nfields, ncol, inp.data_ptr(), out.data_ptr(), _stream()))
        return out

    def pencil_pressure(self, ncol, p_pencil, gp0_pencil):
        _chk(lib().stokes_op_pencil_pressure(self._h, ncol, p_pencil.data_ptr(), gp0_pencil.data_ptr(), _stream()))
        return gp0_pencil

    def mult_schur(self, p, pout, restart=None, rtol=None, atol=1e-50, max_it=10000, inner=None):
        """StokesMatMultSchur (stokes.C:523-535) with the built-in inner GMRES on MatVV (KSP defaults unless given), or with
        `inner`: a callable (b, x) on device velocity tensors that solves VV x = b -- the user's KSPSchurVelocity (stokes.C:531),
        handed to the C entry point as its inner_solve callback exactly as the PETSc adapter does."""
        if restart is not None or rtol is not None:
            _chk(lib().stokes_op_set_inner_solver(self._h, 30 if restart is None else restart, 1e-5 if rtol is None else rtol, atol, max_it))
        cb = None
        if inner is not None:
            n = self.velocity_size

            def tramp(ctx, bp, xp, stream):
                try:
                    inner(device_view(bp, n), device_view(xp, n))
                    return 0
                except Exception:
                    import traceback
                    traceback.print_exc()
                    return 5
            cb = Fgmres.APPLY_FN(tramp)
        _chk(lib().stokes_op_mult_schur(self._h, _dev_ptr(p, self.pressure_size), _dev_ptr(pout, self.pressure_size),
                                        C.cast(cb, C.c_void_p) if cb else None, None, _stream()))
        return pout

    def set_inner_reduce(self, group=None):
        """Slab mode: the built-in inner solve of mult_schur works on distributed velocity vectors."""
        self._red = allreduce_trampoline(group)
        _chk(lib().stokes_op_set_inner_reduce(self._h, C.cast(self._red, C.c_void_p), None))

    @property
    def inner_iterations(self):
        return lib().stokes_op_inner_iterations(self._h)

    def viscosity_range(self):
        """(min, max) of eta after the last `function`: what StokesFunction prints (stokes.C:731-734)."""
        lo, hi = C.c_double(), C.c_double()
        _chk(lib().stokes_op_viscosity_range(self._h, C.byref(lo), C.byref(hi), _stream()))
        return lo.value, hi.value

    def write_vtk(self, state, path):
        """StokesStateView (stokes.C:1821-1894): the -output_vtk dump of a state vector (device tensor)."""
        _chk(lib().stokes_op_write_vtk(self._h, _dev_ptr(state, self.global_size), str(path).encode()))

    def function(self, x, y):
        return self._call(lib().stokes_op_function, x, self.global_size, y, self.global_size)

    def get_state(self, which):
        import numpy as np
        n = self.local_nodes if which < 2 else self.local_nodes * self.d
        out = np.empty(n)
        _chk(lib().stokes_op_get_state(self._h, which, _np_dp(out)))
        return out

    def set_state(self, which, values):
        import numpy as np
        values = np.ascontiguousarray(values, dtype=np.float64)
        n = self.local_nodes if which < 2 else self.local_nodes * self.d
        assert values.size == n
        _chk(lib().stokes_op_set_state(self._h, which, _np_dp(values)))

    def destroy(self):
        if self._h:
            if self._owned:
                lib().stokes_op_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def timers(enable=None, reset=False):
    """Per-stage device timers of the library (chebhip_timers_*): timers(True) switches them on, timers() returns
    {stage name: (total ms, calls)} for every stage that ran, timers(reset=True) clears the counters."""
    L = lib()
    if enable is not None:
        _chk(L.chebhip_timers_enable(1 if enable else 0))
    if reset:
        _chk(L.chebhip_timers_reset())
    out = {}
    i = 0
    while True:
        name = L.chebhip_stage_name(i).decode()
        if not name:
            break
        ms, calls = C.c_double(), C.c_long()
        _chk(L.chebhip_timers_read(i, C.byref(ms), C.byref(calls)))
        if calls.value:
            out[name] = (ms.value, calls.value)
        i += 1
    return out


class FdPc:
    """The finite-difference preconditioner of the reference on the device: FormJacobian's matrix P
    (elliptic.C:537-590) for an EllipticOp, MatVVPC (stokes.C:1160-1241) on velocity vectors for a StokesOp.
    `apply` is an approximate solve with P (fast diagonalisation + `sweeps` defect corrections); pass the object as
    the `M` of Fgmres.solve."""

    def __init__(self, op, sweeps=1, handle=None):
        """handle: a slab-mode handle owned by a slab driver (dist.py: DistStokesC.pc / DistEllipticC.pc) -- borrowed."""
        kind = type(op).__name__
        self._owned = handle is None
        if handle is None:
            h = C.c_void_p()
            _chk((lib().ell_pc_create if kind == "EllipticOp" else lib().stokes_pc_create)(op._h, C.byref(h)))
        else:
            h = handle
        self._h = h
        self._op = op                     # the handle reads the operator's state: keep it alive
        self.n = op.global_size if kind == "EllipticOp" else op.velocity_size
        _chk(lib().chebhip_fdpc_set_sweeps(h, sweeps))

    def update(self):
        """FormJacobian / StokesPCSetUp0: re-assemble from the operator's current eta, deta (gradu)."""
        _chk(lib().chebhip_fdpc_update(self._h, _stream()))

    def mult(self, x, y):
        _chk(lib().chebhip_fdpc_mult(self._h, _dev_ptr(x, self.n), _dev_ptr(y, self.n), _stream()))
        return y

    def apply(self, r, z):
        _chk(lib().chebhip_fdpc_apply(self._h, _dev_ptr(r, self.n), _dev_ptr(z, self.n), _stream()))
        return z

    def apply_cm(self, r, z):
        """MatVVPC solve on component-major velocity vectors (StokesOp.mult_vv_cm), sweeps = 0."""
        _chk(lib().chebhip_fdpc_apply_cm(self._h, _dev_ptr(r, self.n), _dev_ptr(z, self.n), _stream()))
        return z

    def destroy(self):
        if getattr(self, "_h", None):
            if self._owned:
                lib().chebhip_fdpc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class StokesSaddlePc:
    """StokesPCApply0..3 (stokes.C:1714-1817) on the device: block LU / upper / diagonal / lower preconditioners of the
    saddle-point system, with the inner solves KSPVelocity, KSPSchur, KSPSchurVelocity (stokes.C:328-341).
    Pass the object as the `M` of Fgmres.solve around StokesOp.mult."""

    def __init__(self, op, saddle_type=0, vel=(4, 1e-5), schur=(3, 1e-5), svel=(0, 1e-5), pc_sweeps=0, schur_jacobi=True, slab=None):
        """schur_jacobi: KSPSchur's PCJACOBI with 1/eta on the diagonal (stokes.C:330-331, 538-553); False = -schur_pc_type none.
        slab = (slab-mode FdPc, reduce_fn, reduce_ctx): `op` is the slab-mode operator of a slab driver (dist.py); the inner solves
        and the pressure mean complete their sums over the ranks, and apply() is collective."""
        h = C.c_void_p()
        if slab is None:
            _chk(lib().stokes_saddle_create(op._h, C.byref(h)))
        else:
            pc, rfn, rctx = slab
            self._slab_pc = pc
            _chk(lib().stokes_saddle_create_slab(op._h, pc._h, rfn, rctx, C.byref(h)))
        self._h = h
        self._op = op
        self.n = op.global_size
        _chk(lib().stokes_saddle_set_type(h, saddle_type))
        _chk(lib().stokes_saddle_set_pc_sweeps(h, pc_sweeps))
        _chk(lib().stokes_saddle_set_schur_jacobi(h, 1 if schur_jacobi else 0))
        for which, (m, rtol) in enumerate((vel, schur, svel)):
            _chk(lib().stokes_saddle_set_inner(h, which, m, rtol))

    def setup(self):
        """StokesPCSetUp0: call after StokesOp.function has changed the viscosity."""
        _chk(lib().stokes_saddle_setup(self._h, _stream()))

    def apply(self, x, y):
        _chk(lib().stokes_saddle_apply(self._h, _dev_ptr(x, self.n), _dev_ptr(y, self.n), _stream()))
        return y

    inner_iterations = property(lambda self: (lib().stokes_saddle_iterations(self._h, 0), lib().stokes_saddle_iterations(self._h, 1)))

    def destroy(self):
        if getattr(self, "_h", None):
            lib().stokes_saddle_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Fgmres:
    """Restarted flexible GMRES on device vectors (KSPFGMRES's role, elliptic.C:181-185).

    `A` and the optional right preconditioner `M` are operator objects of this module (EllipticOp,
    StokesOp): their C entry points are handed to the solver directly, no Python in the loop.
    """

    def __init__(self, n, restart=30, rtol=1e-5, atol=1e-50, max_it=10000):
        self.n = int(n)
        h = C.c_void_p()
        _chk(lib().chebhip_fgmres_create(self.n, restart, C.byref(h)))
        self._h = h
        _chk(lib().chebhip_fgmres_set_tolerances(h, rtol, atol, max_it))

    APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)

    def _fn(self, op, entry):
        """C entry point + handle of an operator object, or a trampoline around a Python callable (x, y) on
        device tensors (used by the multi-rank drivers of dist.py, whose matvec includes the exchanges)."""
        if op is None:
            return None, None
        kind = type(op).__name__
        if kind == "FdPc":
            return C.cast(lib().chebhip_fdpc_apply, C.c_void_p), op._h
        if kind == "StokesSaddlePc":
            return C.cast(lib().stokes_saddle_apply, C.c_void_p), op._h
        if kind in ("EllipticOp", "StokesOp"):
            name = {"EllipticOp": "ell_op_", "StokesOp": "stokes_op_"}[kind] + entry
            return C.cast(getattr(lib(), name), C.c_void_p), op._h
        n = self.n

        def tramp(ctx, xp, yp, stream):
            try:
                op(device_view(xp, n), device_view(yp, n))
                return 0
            except Exception:
                import traceback
                traceback.print_exc()
                return 5
        cb = Fgmres.APPLY_FN(tramp)
        self._keep.append(cb)
        return C.cast(cb, C.c_void_p), None

    def set_reduce_raw(self, fn, ctx):
        """The same with a chebhip_reduce_fn given as (function pointer, context): Comm.reduce_fn() of dist.py."""
        _chk(lib().chebhip_fgmres_set_reduce(self._h, fn, ctx))

    def set_reduce(self, group=None):
        """Vectors are distributed over the ranks of `group`: complete every inner product with an all-reduce."""
        self._red = allreduce_trampoline(group)
        _chk(lib().chebhip_fgmres_set_reduce(self._h, C.cast(self._red, C.c_void_p), None))

    def solve(self, A, b, x, M=None, x_nonzero=False, a_entry="mult", m_entry="mult"):
        self._keep = []
        fa, ca = self._fn(A, a_entry)
        fm, cm = self._fn(M, m_entry)
        _chk(lib().chebhip_fgmres_solve(self._h, fa, ca, fm, cm, _dev_ptr(b, self.n), _dev_ptr(x, self.n),
                                        1 if x_nonzero else 0, _stream()))
        return x

    iterations = property(lambda self: lib().chebhip_fgmres_iterations(self._h))
    residual = property(lambda self: lib().chebhip_fgmres_residual(self._h))
    reason = property(lambda self: lib().chebhip_fgmres_reason(self._h))

    def destroy(self):
        if getattr(self, "_h", None):
            lib().chebhip_fgmres_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass
