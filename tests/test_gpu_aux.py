"""Auxiliary pieces around the hot path: per-stage device timers, the viscosity range StokesFunction
prints (stokes.C:731-734) and the -output_vtk dump StokesStateView (stokes.C:1821-1894)."""
import re
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
import oracle_lib as orc
from conftest import relerr

pytestmark = pytest.mark.gpu
sp = ge.load()
SEED = 20240229
POWER = (1, 1.0, 3.0, 1e-4, 1.0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).cuda()


def test_stage_timers():
    op = sp.EllipticOp((40, 36, 32))
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
    assert sp.timers(enable=False, reset=True) == {}
    op.mult(U, V)                                     # timers off: nothing recorded
    assert sp.timers() == {}
    sp.timers(enable=True)
    for _ in range(7):
        op.mult(U, V)
    ks = sp.Fgmres(op.global_size, restart=10, rtol=1e-3, max_it=10)
    ks.solve(op, V, U)
    t = sp.timers(enable=False)
    # the 7 direct applies plus the solver's (one more may have been enqueued speculatively)
    assert 7 + ks.iterations <= t["ell_op_mult"][1] <= 7 + ks.iterations + 2
    assert t["ell_op_mult"][0] > 0 and t["chebhip_fgmres_solve"][1] == 1 and t["chebhip_fgmres_solve"][0] >= 0
    sp.timers(reset=True)
    assert sp.timers() == {}
    ks.destroy(); op.destroy()


@pytest.mark.parametrize("dims", [(14, 12), (10, 9, 8)], ids=lambda d: "x".join(map(str, d)))
def test_viscosity_range(dims):
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 1)
    rng = np.random.default_rng(SEED)
    xs = U + 0.05 * rng.standard_normal(U.shape)
    st.set_rheology(*POWER); st.set_dirichlet(dv); st.set_force(U2)
    y = torch.empty(st.global_size, dtype=torch.float64, device="cuda")
    st.function(dev(xs), y)
    lo, hi = st.viscosity_range()
    _, eta, _, _ = orc.stokes_function(dims, xs, dv, U2, POWER)
    assert abs(lo - eta.min()) <= 1e-10 * eta.min() and abs(hi - eta.max()) <= 1e-10 * eta.max()
    st.destroy()


def _sections(path):
    """{name: array} of a legacy VTK file as StokesStateView writes it."""
    txt = open(path).read()
    assert txt.startswith("# vtk DataFile Version 2.0\nStokes Output\nASCII\nDATASET STRUCTURED_GRID\n")
    out = {}
    m = re.search(r"DIMENSIONS (\d+) (\d+) (\d+)\nPOINTS (\d+) double\n", txt)
    out["dims"] = tuple(int(v) for v in m.groups()[:3]); n = int(m.group(4))
    heads = [("POINTS", r"POINTS \d+ double\n", 3), ("velocity", r"VECTORS velocity double\n", 3), ("pressure", r"SCALARS pressure double 1\nLOOKUP_TABLE default\n", 1),
             ("vel_force", r"VECTORS vel_force double\n", 3), ("div_force", r"SCALARS div_force double 1\nLOOKUP_TABLE default\n", 1),
             ("eta", r"SCALARS eta double 1\nLOOKUP_TABLE default\n", 1), ("deta", r"SCALARS deta double 1\nLOOKUP_TABLE default\n", 1),
             ("strain", r"TENSORS strain double\n", 9)]
    for name, pat, per in heads:
        mm = re.search(pat, txt)
        vals = txt[mm.end():].split()[:n * per]
        out[name] = np.array([float(v) for v in vals]).reshape(n, per)
    return out


@pytest.mark.parametrize("dims", [(8, 7), (7, 6, 5)], ids=lambda d: "x".join(map(str, d)))
def test_vtk_dump(dims, tmp_path):
    d = len(dims)
    st = sp.StokesOp(dims)
    U, U2, dv = orc.stokes_exact(dims, 1)
    rng = np.random.default_rng(SEED)
    xs = U + 0.05 * rng.standard_normal(U.shape)
    st.set_rheology(*POWER); st.set_dirichlet(dv); st.set_force(U2)
    xd = dev(xs); y = torch.empty_like(xd)
    st.function(xd, y)
    path = tmp_path / "stokes.vtk"
    st.write_vtk(xd, path)
    s = _sections(path)
    N = int(np.prod(dims))
    assert s["dims"] == (dims[0], dims[1], dims[2] if d > 2 else 1) and s["POINTS"].shape[0] == N
    # coordinates x = cos(i pi/(dim-1)) (stokes.C:296), row-major with the last dimension fastest
    grids = np.meshgrid(*[np.cos(np.arange(p) * np.pi / (p - 1)) for p in dims], indexing="ij")
    for j in range(d):
        assert np.abs(s["POINTS"][:, j] - grids[j].ravel()).max() < 1e-6
    _, eta, deta, strain = orc.stokes_function(dims, xs, dv, U2, POWER, mode=orc.DIRECT)
    assert relerr(s["eta"][:, 0], eta) < 1e-6 and relerr(s["deta"][:, 0], deta) < 1e-6      # "%20e": 7 significant digits
    # velocity = interior values of the state + Dirichlet values; pressure = interior + extrapolated boundary
    m = np.ones(dims, dtype=bool)
    for ax, p in enumerate(dims):
        sl = [slice(None)] * d; sl[ax] = 0; m[tuple(sl)] = False; sl[ax] = p - 1; m[tuple(sl)] = False
    X = xs.reshape(-1, d + 1)
    vel = np.zeros((N, d)); vel[m.ravel()] = X[:, :d]; vel[~m.ravel()] = dv.reshape(-1, d)
    assert relerr(s["velocity"][:, :d], vel) < 1e-6
    pL = np.zeros(N); pL[m.ravel()] = X[:, d]
    assert relerr(s["pressure"][:, 0], orc.stokes_pressure_reduce(dims, pL)) < 1e-6
    T = s["strain"].reshape(N, 3, 3)
    ref = strain.reshape(d, N, d)                       # strain[j][i*d+k]
    for j in range(d):
        for k in range(d):
            assert np.abs(T[:, j, k] - ref[j][:, k]).max() <= 1e-6 * max(1.0, np.abs(ref).max())
    st.destroy()


def test_results_are_bitwise_reproducible():
    """Every kernel sums in a fixed order (no atomics on data): the same call on the same input gives the same bits -- the
    Poisson matvec, FormFunction + Jacobian apply, StokesMatMult, the preconditioner and a whole FGMRES solve."""
    import numpy as np
    import torch
    torch.manual_seed(3)
    op = sp.EllipticOp((132, 70, 68))
    u = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5
    b = torch.randn_like(u)
    outs = []
    for _ in range(2):
        v = torch.empty_like(u); r = torch.empty_like(u); j = torch.empty_like(u); z = torch.empty_like(u); x = torch.zeros_like(u)
        lin = sp.EllipticOp((132, 70, 68)); lin.mult(b, v); lin.destroy()
        op.set_dirichlet(np.full(op.dirichlet_size, 0.25))
        op.function(u, b, r, 2.0, 2.0); op.mult(b, j)
        pc = sp.FdPc(op, sweeps=0); pc.apply(b, z)
        ks = sp.Fgmres(op.global_size, restart=10, rtol=1e-6, max_it=10)
        ks.solve(op, b, x, M=pc)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (v, r, j, z, x)])
        pc.destroy()
    for a, c in zip(*outs):
        assert torch.equal(a, c)
    op.destroy()
    st = sp.StokesOp((40, 36, 34))
    st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size)); st.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    xs = torch.randn(st.global_size, dtype=torch.float64, device="cuda")
    ys = []
    for _ in range(2):
        f = torch.empty_like(xs); m = torch.empty_like(xs)
        st.function(xs, f); st.mult(xs, m); torch.cuda.synchronize()
        ys.append((f.clone(), m.clone()))
    assert torch.equal(ys[0][0], ys[1][0]) and torch.equal(ys[0][1], ys[1][1])
    st.destroy()


def test_calls_on_a_non_default_stream():
    """Every entry point takes the caller's stream: results on a side stream (with the default stream kept busy) equal those on
    the default stream -- the Stokes pressure chain forks to the operator's own second stream and must join the caller's."""
    import numpy as np
    import torch
    torch.manual_seed(11)
    st = sp.StokesOp((48, 40, 36))
    st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size)); st.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    op = sp.EllipticOp((132, 70, 68)); op.set_dirichlet(np.full(op.dirichlet_size, 0.5))
    xs = torch.randn(st.global_size, dtype=torch.float64, device="cuda")
    u = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5
    ref = [torch.empty_like(xs), torch.empty_like(xs), torch.empty_like(u), torch.empty_like(u)]
    st.function(xs, ref[0]); st.mult(xs, ref[1]); op.function(u, None, ref[2], 1.0, 2.0); op.mult(u, ref[3])
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    out = [torch.full_like(r, float("nan")) for r in ref]
    busy = torch.randn(4096, 4096, device="cuda")
    for _ in range(3):
        busy = busy @ busy * 1e-4                        # the default stream has work queued while the side stream runs
    with torch.cuda.stream(side):
        st.function(xs, out[0]); st.mult(xs, out[1]); op.function(u, None, out[2], 1.0, 2.0); op.mult(u, out[3])
    side.synchronize()
    for a, b in zip(out, ref):
        assert torch.equal(a, b)
    torch.cuda.synchronize()
    st.destroy(); op.destroy()


def test_first_call_on_a_non_blocking_stream_while_the_null_stream_is_busy():
    """State the library allocates on first use (w0, eta, gradu of the elliptic callbacks) is cleared on the null stream,
    which a non-blocking stream does not wait for: the allocating call must, or the clear lands on top of what the caller's
    stream has written meanwhile.  Here the null stream is kept busy (torch's default stream IS the null stream) so that a
    missing wait shows every time: the clear would land after FormFunction has stored its linearisation state, and the
    Jacobian apply that follows would use zeros.  (Found by tools/fuzz_dist_threads.py, whose thread ranks each own a
    non-blocking stream; there the clear landed between two kernels once in a few hundred cases.)"""
    import numpy as np
    import torch
    dims = (40, 36, 34)

    def run(side):
        op = sp.EllipticOp(dims); op.set_dirichlet(np.full(op.dirichlet_size, 0.5))
        g = torch.Generator(device="cuda").manual_seed(5)
        u = torch.rand(op.global_size, dtype=torch.float64, device="cuda", generator=g) + 0.5
        outs = [torch.full_like(u, float("nan")) for _ in range(3)]
        torch.cuda.synchronize()
        if side is not None:
            torch.cuda._sleep(400_000_000)                  # ~0.2 s on the null stream, queued ahead of anything the library puts there
        with torch.cuda.stream(side if side is not None else torch.cuda.current_stream()):
            op.function(u, None, outs[0], 1.5, 3.0); op.mult(u, outs[1])
            torch.cuda.current_stream().synchronize()
        torch.cuda.synchronize()                            # the null stream has drained: whatever was queued there has landed
        state = [op.get_state(w) for w in range(2 + len(dims))]      # eta, deta, gradu[k] as FormFunction left them
        op.destroy()
        return outs[:2], state
    got, got_state = run(torch.cuda.Stream())
    ref, ref_state = run(None)
    for a, b in zip(got, ref):
        assert torch.isfinite(a).all() and torch.equal(a, b)
    for w, (a, b) in enumerate(zip(got_state, ref_state)):
        assert np.array_equal(a, b), "state array %d was overwritten after FormFunction stored it" % w
    assert np.abs(ref_state[2]).max() > 0 and ref_state[0].min() > 1.0


def test_get_state_after_formfunction_still_queued_on_a_non_blocking_stream():
    """The interior-line FormFunction leaves eta / eta' to be rebuilt from w0 on demand (ell_sync_coeffs, a null-stream kernel).
    ell_op_get_state must wait for a FormFunction that is still queued on the caller's non-blocking stream BEFORE that rebuild
    reads w0 (round-3 advisor finding): the side stream is held up by a sleep so that a missing wait shows every time."""
    import numpy as np
    import torch
    dims = (72, 68, 66)                                     # the interior-line path: even extents of 66..256, homogeneous rows, exponent 2
    op = sp.EllipticOp(dims)
    g = torch.Generator(device="cuda").manual_seed(9)
    u = torch.rand(op.global_size, dtype=torch.float64, device="cuda", generator=g) + 0.5
    r = torch.empty_like(u)
    op.function(u, None, r, 1.0, 2.0); torch.cuda.synchronize()      # allocate the state; a first, different coefficient state
    u2 = u * 1.5
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(400_000_000)                      # ~0.2 s: FormFunction below is still queued when get_state is called
        op.function(u2, None, r, 4.0, 2.0)
    eta = op.get_state(0); deta = op.get_state(1)           # no synchronisation by the caller
    torch.cuda.synchronize()
    interior = np.zeros(dims); interior[1:-1, 1:-1, 1:-1] = u2.cpu().numpy().reshape([n - 2 for n in dims])
    assert np.allclose(eta.reshape(dims), 1.0 + 4.0 * interior ** 2, rtol=1e-14, atol=0)
    assert np.allclose(deta.reshape(dims), 2.0 * 4.0 * interior, rtol=1e-14, atol=0)
    op.destroy()


def test_two_host_threads_with_their_own_handles():
    """One handle belongs to one host thread at a time, but two threads may drive two handles at once (ctypes releases the GIL):
    the library's shared state -- lazily read environment switches, the CU-count cache, the timer registry, rocBLAS handles --
    must not couple them."""
    import threading
    import numpy as np
    import torch
    results, errors = {}, []

    def work(tag, dims, seed):
        try:
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                op = sp.EllipticOp(dims)
                g = torch.Generator(device="cuda").manual_seed(seed)
                U = torch.randn(op.global_size, dtype=torch.float64, device="cuda", generator=g)
                V = torch.empty_like(U); acc = torch.zeros_like(U)
                for _ in range(40):
                    op.mult(U, V); acc += V
                s.synchronize()
                results[tag] = (U.cpu().numpy(), (acc / 40).cpu().numpy(), dims)
                op.destroy()
        except Exception as e:                              # surfaced in the main thread
            errors.append(repr(e))
    ts = [threading.Thread(target=work, args=("a", (132, 70, 68), 1)), threading.Thread(target=work, args=("b", (40, 300), 2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for tag in ("a", "b"):
        U, V, dims = results[tag]
        assert relerr(V, orc.elliptic_mult(dims, U, mode=orc.FAST, nthreads=8)) < 1e-10


def test_options_and_kernel_families_agree():
    """chebhip_set_option (the library's only switches; it never reads the environment): unknown names fail, values read
    back, and the kernel families an option selects give the same operator: the general 8-byte sweep kernel against the
    16-byte kernels at 96^3, separate launches against the multi-job launch of the Stokes sweeps, FormFunction with and
    without its gather pass."""
    import ctypes as C
    import numpy as np
    L = sp.lib()
    assert L.chebhip_set_option(b"no_such_option", 1) == 4 and b"unknown option" in L.chebhip_last_error()
    names = sp.options()
    assert {"general_kernels", "separate_launches", "gather_pass", "local_timeout_s"} <= set(names) and names["local_timeout_s"] == 120
    sp.set_option("equal_shares", 1); assert sp.get_option("equal_shares") == 1; sp.set_option("equal_shares", 0)
    dims = (96, 96, 96)
    op = sp.EllipticOp(dims)
    U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    b = torch.randn_like(U); us = torch.rand_like(U) + 0.5
    V0, V1, R0, R1 = (torch.empty_like(U) for _ in range(4))
    op.mult(U, V0)                                       # linear state: the D D kernels
    try:
        sp.set_option("general_kernels", 1)
        op.mult(U, V1)
    finally:
        sp.set_option("general_kernels", 0)
    op.function(us, b, R0, 1.5, 2.0)                     # interior-line path, eta on chip
    J0, J1 = torch.empty_like(U), torch.empty_like(U)
    op.mult(U, J0)                                       # Jacobian apply from that state
    try:
        sp.set_option("gather_pass", 1); sp.set_option("eta_from_memory", 1)
        op.function(us, b, R1, 1.5, 2.0)
        op.mult(U, J1)
    finally:
        sp.set_option("gather_pass", 0); sp.set_option("eta_from_memory", 0)
    torch.cuda.synchronize()
    assert float((V1 - V0).norm() / V0.norm()) < 1e-12 and float((R1 - R0).norm() / R0.norm()) < 1e-12
    assert float((J1 - J0).norm() / J0.norm()) < 1e-12
    op.destroy()
    st = sp.StokesOp((48, 40, 36))
    x = torch.randn(st.global_size, dtype=torch.float64, device="cuda"); y0, y1 = torch.empty_like(x), torch.empty_like(x)
    st.mult(x, y0)
    try:
        sp.set_option("separate_launches", 1)
        st.mult(x, y1)
    finally:
        sp.set_option("separate_launches", 0)
    torch.cuda.synchronize()
    assert float((y1 - y0).norm() / y0.norm()) < 1e-13
    st.destroy()
    # "general_kernels" covers the Stokes launches too (round-3 advisor finding): a handle created and driven with the option set
    # keeps all nine stress components (the six-component storage feeds spaced-out input fields only the 16-byte kernels
    # read) and runs every sweep of its multi-job launches on the general kernel; same operator to rounding
    dims = (68, 66, 66)                                  # even extents above 64: the six-component storage by default
    g = torch.Generator(device="cuda").manual_seed(13)
    outs = []
    for general in (0, 1):
        sp.set_option("general_kernels", general)
        try:
            st = sp.StokesOp(dims); st.set_rheology(1, 1.0, 3.0, 1e-3, 1.0)
            st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size))
            if not outs:
                x = torch.randn(st.global_size, dtype=torch.float64, device="cuda", generator=g)
            f, m = torch.empty_like(x), torch.empty_like(x)
            st.function(x, f); st.mult(x, m); torch.cuda.synchronize()
            outs.append((f.clone(), m.clone()))
            st.destroy()
        finally:
            sp.set_option("general_kernels", 0)
    for a_, b_ in zip(outs[0], outs[1]):
        assert float((a_ - b_).norm() / a_.norm()) < 1e-12
    # the constant-coefficient matvec: one launch of d jobs + a sum in the chain's order (1), a launch per direction (2), and the default by
    # size (0: below 6 M unknowns in 3-D two jobs + a last direction that adds both terms as it stores, OUT_ACC2): same bits.  Short
    # and long lines, mixed KS (the default then falls back), both tilings of the last direction
    for dims in ((66, 68, 72), (40, 36, 34), (130, 72), (33, 17, 9), (130, 128, 126), (98, 100, 130), (20, 18, 16), (64, 64, 64)):
        op = sp.EllipticOp(dims)
        U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); Va, Vb, Vc = torch.empty_like(U), torch.empty_like(U), torch.empty_like(U)
        try:
            sp.set_option("poisson_launches", 1); op.mult(U, Va)
            sp.set_option("poisson_launches", 2); op.mult(U, Vb)
            sp.set_option("poisson_launches", 0); op.mult(U, Vc)
        finally:
            sp.set_option("poisson_launches", 0)
        torch.cuda.synchronize()
        assert torch.equal(Va, Vb) and torch.equal(Va, Vc), dims
        op.destroy()
    # large 3-D grids (6 M unknowns and more, round 6): two jobs in one launch + a last direction with two operands (3: always; 0: while the
    # padded field has at most 9 M values) against a launch per direction (2): same bits.  256-point lines (ONE operand set in the kernel),
    # 200-point lines, a last direction of 128 interior points (the operand pair of the 128-point kernel), unequal first extents (the
    # jobs cannot share a launch: the route falls back)
    for dims in ((256, 256, 256), (200, 200, 200), (256, 256, 130), (256, 130, 256)):
        op = sp.EllipticOp(dims)
        U = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); Va, Vb, Vc = torch.empty_like(U), torch.empty_like(U), torch.empty_like(U)
        try:
            sp.set_option("poisson_launches", 0); op.mult(U, Va); op.mult(U, Va)
            sp.set_option("poisson_launches", 2); op.mult(U, Vb)
            sp.set_option("poisson_launches", 3); op.mult(U, Vc); op.mult(U, Vc)
        finally:
            sp.set_option("poisson_launches", 0)
        torch.cuda.synchronize()
        assert torch.equal(Va, Vb) and torch.equal(Vc, Vb), dims
        op.destroy()
    # the pressure gradient with each direction's end-point extrapolation folded into its matrix (the default) against the
    # three extrapolation passes of StokesPressureReduceOrder followed by plain D (option pressure_passes, read at create)
    for dims in ((48, 40, 36), (30, 41), (66, 68, 72)):
        outs = []
        for passes in (0, 1):
            sp.set_option("pressure_passes", passes)
            try:
                st = sp.StokesOp(dims)
            finally:
                sp.set_option("pressure_passes", 0)
            st.set_rheology(1, 1.0, 3.0, 1e-3, 1.0)
            g = torch.Generator(device="cuda").manual_seed(9)
            x = torch.randn(st.global_size, dtype=torch.float64, device="cuda", generator=g)
            p = torch.randn(st.pressure_size, dtype=torch.float64, device="cuda", generator=g)
            f, m, vp = torch.empty_like(x), torch.empty_like(x), torch.empty(st.velocity_size, dtype=torch.float64, device="cuda")
            st.function(x, f); st.mult(x, m); st.mult_vp(p, vp)
            torch.cuda.synchronize()
            outs.append((f, m, vp)); st.destroy()
        for a, b in zip(*outs):
            assert float((a - b).norm() / b.norm()) < 1e-12, dims


@pytest.mark.parametrize("dims", [(48, 40, 36), (30, 41), (66, 68, 72), (33, 17, 9)], ids=lambda d: "x".join(map(str, d)))
def test_uniform_viscosity_jacobian_path(dims):
    """With a uniform viscosity and eta' = 0 (after create, after a StokesFunction with the linear rheology, after
    set_state(eta = const)) StokesMatMult / StokesMatMultVV run -eta/2 (sum_j D_j D_j v + grad div v) instead of the node loop
    (option general_viscous, read at create, keeps the general block): both against each other and against the oracle; a
    non-uniform viscosity or a power-law state must take the general path again."""
    d = len(dims)
    rng = np.random.default_rng(SEED)
    ops = []
    for general in (0, 1):
        sp.set_option("general_viscous", general)
        try:
            ops.append(sp.StokesOp(dims))
        finally:
            sp.set_option("general_viscous", 0)
    fast, gen = ops
    x = rng.standard_normal(fast.global_size); v = rng.standard_normal(fast.velocity_size)
    N = fast.local_nodes

    def both(state):
        outs = []
        for op in ops:
            state(op)
            y = torch.empty(op.global_size, dtype=torch.float64, device="cuda"); yv = torch.empty(op.velocity_size, dtype=torch.float64, device="cuda")
            op.mult(dev(x), y); op.mult_vv(dev(v), yv); torch.cuda.synchronize()
            outs.append((y.cpu().numpy(), yv.cpu().numpy()))
        return outs
    # 1. the state after create: eta = 1
    (y0, v0), (y1, v1) = both(lambda op: None)
    assert relerr(y0, y1) < 1e-12 and relerr(v0, v1) < 1e-12
    assert relerr(y0, orc.stokes_mult(dims, x, mode=orc.DIRECT)) < 1e-10 and relerr(v0, orc.stokes_mult_vv(dims, v, mode=orc.DIRECT)) < 1e-10
    # 2. a uniform viscosity other than 1
    (y0, v0), (y1, v1) = both(lambda op: op.set_state(0, np.full(N, 2.5)))
    assert relerr(y0, y1) < 1e-12 and relerr(v0, v1) < 1e-12
    assert relerr(v0, orc.stokes_mult_vv(dims, v, eta=np.full(N, 2.5), mode=orc.DIRECT)) < 1e-10
    # 3. a variable viscosity: the general path in both handles (bitwise the same code)
    eta = np.exp(rng.uniform(-1, 1, N))
    (y0, v0), (y1, v1) = both(lambda op: op.set_state(0, eta))
    assert np.array_equal(y0, y1) and np.array_equal(v0, v1)
    assert relerr(v0, orc.stokes_mult_vv(dims, v, eta=eta, mode=orc.DIRECT)) < 1e-10
    # 4. StokesFunction: linear rheology re-arms the fast path, power law disarms it
    dv = rng.standard_normal(fast.dirichlet_size); f = rng.standard_normal(fast.global_size)
    for op in ops:
        op.set_dirichlet(dv); op.set_force(f)
    def fn(rheo):
        def go(op):
            op.set_rheology(*rheo); r = torch.empty(op.global_size, dtype=torch.float64, device="cuda"); op.function(dev(x), r)
        return go
    (y0, v0), (y1, v1) = both(fn((0, 1.0, 1.0, 1.0, 1.0)))
    assert relerr(y0, y1) < 1e-12 and relerr(y0, orc.stokes_mult(dims, x, mode=orc.DIRECT)) < 1e-10
    (y0, v0), (y1, v1) = both(fn(POWER))
    assert np.array_equal(y0, y1)
    _, eta_p, deta_p, strain_p = orc.stokes_function(dims, x, dv, f, rheology=POWER, mode=orc.DIRECT)
    assert relerr(y0, orc.stokes_mult(dims, x, eta_p, deta_p, strain_p, mode=orc.DIRECT)) < 1e-10
    for op in ops:
        op.destroy()


@pytest.mark.parametrize("dims", [(48, 40, 36), (30, 41), (66, 68, 72), (33, 17, 9)], ids=lambda d: "x".join(map(str, d)))
def test_linear_stokes_function_on_the_uniform_route(dims):
    """StokesFunction with the linear rheology (stokes.C:1920-1926) runs -1/2 (sum_j D_j D_j v + grad div v) on the velocity with
    its Dirichlet values instead of the node loop; the symmetrised strain it would leave as state (stokes.C:722) is rebuilt
    on demand from the handle's copy of that velocity.  Against the general route (option general_viscous), against the
    oracle, with the state read back after later callbacks have overwritten the work vectors, and with an eta' set by
    hand afterwards (the Jacobian apply must then meet the strain of THAT residual)."""
    d = len(dims)
    rng = np.random.default_rng(SEED + 7)
    ops = []
    for general in (0, 1):
        sp.set_option("general_viscous", general)
        try:
            ops.append(sp.StokesOp(dims))
        finally:
            sp.set_option("general_viscous", 0)
    x = rng.standard_normal(ops[0].global_size); x2 = rng.standard_normal(ops[0].global_size)
    dv = rng.standard_normal(ops[0].dirichlet_size); f = rng.standard_normal(ops[0].global_size)
    N = ops[0].local_nodes
    ref_y, ref_eta, ref_deta, ref_strain = orc.stokes_function(dims, x, dv, f, mode=orc.DIRECT)
    deta_hand = rng.uniform(0.1, 0.5, N)
    ref_j = orc.stokes_mult(dims, x2, eta=ref_eta, deta=deta_hand, strain=ref_strain, mode=orc.DIRECT)
    outs = []
    for op in ops:
        op.set_dirichlet(dv); op.set_force(f)
        op.set_rheology(1, 1.0, 3.0, 1e-3, 1.0)                # a power-law state first: eta, eta' must be reset by the linear call
        y = torch.empty(op.global_size, dtype=torch.float64, device="cuda"); m = torch.empty_like(y)
        op.function(dev(x2), y)
        op.set_rheology(0, 1.0, 1.0, 1.0, 1.0)
        op.function(dev(x), y)
        op.mult(dev(x2), m)                                    # overwrites the work vectors of the callbacks
        torch.cuda.synchronize()
        y = y.cpu().numpy()
        assert relerr(y, ref_y) < 1e-10
        assert np.array_equal(op.get_state(0), np.ones(N)) and np.array_equal(op.get_state(1), np.zeros(N))
        for j in range(d):
            assert relerr(op.get_state(2 + j), ref_strain[j]) < 1e-10
        assert relerr(m.cpu().numpy(), orc.stokes_mult(dims, x2, mode=orc.DIRECT)) < 1e-10
        # a fresh residual, then eta' by hand: the Jacobian apply needs the strain of this residual
        op.function(dev(x), torch.empty_like(m))
        op.set_state(1, deta_hand)
        op.mult(dev(x2), m); torch.cuda.synchronize()
        assert relerr(m.cpu().numpy(), ref_j) < 1e-10
        outs.append(y)
        op.destroy()
    assert relerr(outs[0], outs[1]) < 1e-12


def test_cheb_apply_on_an_array_of_a_gigabyte():
    """Arrays of 0.94 GB and more are beyond the 32-bit buffer offsets of the long-line kernel: the general kernel takes
    them (sweep_vec_eligible).  Checked against the same plan applied to the two halves of the tensor (dimension 0 is not
    the transform dimension, so the halves are independent problems below the limit)."""
    dims = (4, 256, 130048)            # 4 * 256 * 130048 * 8 B = 1.07 GB, lines of 256 points at stride 130048
    n = dims[0] * dims[1] * dims[2]
    x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    big = sp.ChebPlan(dims, 1)
    big.mult(x, y)
    half = sp.ChebPlan((2, dims[1], dims[2]), 1)
    yh = torch.empty(n // 2, dtype=torch.float64, device="cuda")
    for h in range(2):
        half.mult(x[h * (n // 2):(h + 1) * (n // 2)], yh)
        torch.cuda.synchronize()
        ref = yh
        got = y[h * (n // 2):(h + 1) * (n // 2)]
        assert float((got - ref).norm() / ref.norm()) < 1e-13
    big.destroy(); half.destroy()


def test_fast_paths_are_the_ones_that_run():
    """Sweep-launch counts of the BASELINE-size callbacks (chebhip_launch_count: one per sweep-kernel launch, multi-job launches
    count once): a silent fall-back to a slower route -- an eligibility test that stops matching, a lost alignment -- changes them.
      256^3 Poisson matvec                3  (one launch per direction; fields of at most 9 M values take two, profiles/r06_two_launch_ab.txt)
      128^3 Poisson matvec                2  (two jobs + a last direction with a two-operand accumulate)
      64^3 linear StokesMatMult           2  (uniform-viscosity route: nine jobs, then the three sweeps of grad div v)
      128^3 power-law StokesMatMultVV     2  (x / y gradient, x / y divergence; the z direction is k_st_zfused16, not a sweep launch)
      128^3 power-law StokesMatMult       2  (round 5: the pressure-gradient sweeps are jobs of the x / y gradient launch), StokesFunction 2
      MatVVPC solve at 128^3              4  (sweep launches: two forward and two backward line transforms with the pointwise steps
                                             inside; the z direction -- forward, scaling, backward -- is k_fdm_zsolve16, not a sweep launch)"""
    L = sp.lib()
    def count(fn):
        torch.cuda.synchronize(); before = L.chebhip_launch_count(); fn(); torch.cuda.synchronize(); return L.chebhip_launch_count() - before
    rnd = lambda n: torch.randn(n, dtype=torch.float64, device="cuda")
    for P, expect in ((256, 3), (208, 2), (128, 2)):
        op = sp.EllipticOp((P, P, P)); U = rnd(op.global_size); V = torch.empty_like(U)
        op.mult(U, V)
        assert count(lambda: op.mult(U, V)) == expect, P
        op.destroy(); del U, V
    st = sp.StokesOp((64, 64, 64)); st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size))
    x = rnd(st.global_size); y = torch.empty_like(x)
    st.function(x, y)
    assert count(lambda: st.mult(x, y)) == 2
    st.destroy()
    st = sp.StokesOp((128, 128, 128)); st.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size))
    x = rnd(st.global_size); y = torch.empty_like(x); v = rnd(st.velocity_size); w = torch.empty_like(v)
    st.function(x, y)
    assert count(lambda: st.mult_vv(v, w)) == 2
    assert count(lambda: st.mult_vv_cm(v, w)) == 2
    assert count(lambda: st.mult(x, y)) == 2
    assert count(lambda: st.function(x, y)) == 2
    pc = sp.FdPc(st, sweeps=0); pc.apply(v, w)
    assert count(lambda: pc.apply(v, w)) == 4
    assert count(lambda: pc.apply_cm(v, w)) == 4
    pc.destroy(); st.destroy()
