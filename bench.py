#!/usr/bin/env python3
"""bench.py -- spectral matvecs/s of the 3-D Chebyshev Poisson operator apply on MI355X.

One "step" = one MatMult_Elliptic (elliptic.C:297-339) on the -dim P,P,P grid with the linear
Poisson state (gamma = 0: eta == 1, deta == 0), global in/out vectors resident in HBM.

  python bench.py --gpus 1 --steps K --warmup W            (N = 1: whole 256^3 grid on one GPU)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
                                                           (N > 1: the same 256^3 grid slab-split
                                                            over N ranks, RCCL all-to-all transposes;
                                                            strong scaling)
  python bench.py --gpus N ...                             (the same, typed without a launcher: bench.py starts
                                                            the N ranks itself as child processes -- before it
                                                            touches the GPU -- and exits with their status)
Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (cheb_sweep_vec4_kernel,
csrc/sweep_vec.hip) with the algorithmic bytes of SURVEY 8(d): 112 B/point per matvec (the six-ChebMult
model), spread over the 3 launches that carry it.  That figure is a MODEL figure of merit: the
constant-coefficient path moves fewer bytes than the model (one launch per direction instead of two sweeps),
so `roofline.frac` is not HBM utilisation.  `roofline.traffic` holds the bytes a launch really moved (PMC
counters, profiles/) when the committed measurement belongs to the sources being timed, and
`roofline.hbm_real_frac` = traffic / launch time / 8 TB/s; `roofline.mfma_f64_frac` is the fraction of the
FP64 matrix peak that actually bounds the kernel at P = 256.  `cpu_baseline` times the CPU oracle (a port of
the reference's pass structure; FFTW/PETSc are not installed) on this box's host cores, rank 0, N = 1 only.
`spinup` untimed matvecs run before the W warm-up steps (setup: an idle MI355X needs a few ms of load before
its clocks settle) and are reported in the line.  After the timed region (N = 1, informational): `extras_us` (the other
callbacks of BASELINE configs 2, 4, 5), `dist_rank_compute` (one slab rank's kernels without a wire) and `solves` (end-to-end
solves of the callers; each is run twice and the second run is `seconds`, the first `seconds_first_run`).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK = 8.0e12          # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_MFMA_PEAK = 78.6e12   # flop/s, v_mfma_f64_16x16x4_f64 dense peak (= FP64 vector peak)
BYTES_PER_POINT = 112.0    # six-sweep Poisson matvec
SEED = 20240229


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--spinup", type=int, default=100, help="untimed setup matvecs before the W warm-up steps (reported as `spinup`)")
    ap.add_argument("--size", type=int, default=256, help="points per dimension P (BASELINE: 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary timings of the other BASELINE configs")
    ap.add_argument("--cpu-threads", type=int, default=1)
    ap.add_argument("--launch-selftest", action="store_true",
                    help="N > 1 plumbing only (tests/test_bench_launch.py): spawn, rendezvous, one all-reduce, rank 0's line; no GPU, no compute, no metric")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` typed as it stands (no launcher, WORLD_SIZE unset): start the N ranks ourselves.  The parent has
    not imported torch.cuda nor touched the GPU in any way (on this pool a process that has initialised the GPU must never exec,
    and a parent holding the GPU would be a rank too many on the card): it starts `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port> bench.py <the same arguments>` as a CHILD process
    (subprocess, not exec), whose rank 0 prints the JSON line on the stdout it inherits, and exits with the child's status."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL between processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // args.gpus)))
    env["BENCH_SPAWNED_BY"] = str(os.getpid())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def launch_selftest(args, world, rank):
    """--launch-selftest: what the self-spawning entry adds, and nothing else -- the ranks exist, meet on 127.0.0.1 over
    BENCH_DIST_BACKEND (gloo on CPU), agree on one all-reduce, and rank 0's single JSON line reaches the caller's stdout."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("BENCH_DIST_BACKEND", "gloo")
    if backend == "nccl":
        raise SystemExit("--launch-selftest is a CPU check of the launcher: BENCH_DIST_BACKEND=gloo")
    wd = Watchdog(rank)
    wd.enter("--launch-selftest: rendezvous and one all-reduce", int(os.environ.get("BENCH_WATCHDOG_S", "600")))
    if world > 1:
        dist.init_process_group(backend=backend)
    if os.environ.get("BENCH_SELFTEST_HANG_RANK", "") == str(rank):     # tests/test_bench_launch.py: a rank that never joins the collective
        time.sleep(1e6)
    t = torch.tensor([rank + 1], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t)
    wd.leave()
    if rank == 0:
        print(json.dumps({"launch_selftest": True, "n_gpus": world, "rank_sum": int(t.item()), "backend": backend,
                          "spawned_by_bench": "BENCH_SPAWNED_BY" in os.environ}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


class Watchdog:
    """N > 1 only: a rank that sits in a collective its peers never join (first RCCL contact, a peer that died) would hang until the
    caller's own limit kills it without a trace.  After `seconds` in one phase this prints which phase it was and ends the process
    with status 3 -- never 0: a hang must not read as success.  (No re-exec, no signal: os._exit from a timer thread.)"""

    def __init__(self, rank):
        self.rank, self.timer, self.phase = rank, None, None

    def enter(self, phase, seconds):
        import threading
        self.leave()
        self.phase = phase

        def fire():
            print("bench.py: WATCHDOG -- rank %d: %s did not return within %d s; ending this rank with status 3" % (self.rank, phase, seconds),
                  file=sys.stderr, flush=True)
            os._exit(3)
        self.timer = threading.Timer(float(seconds), fire)
        self.timer.daemon = True
        self.timer.start()

    def leave(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def roofline_record(P, world, steps, dev_ms, launches_per_step, traffic):
    """The `roofline` object of the line.
    N = 1: the dominant kernel is cheb_sweep_vec4_kernel, 3 launches per matvec, one per direction (each replaces 2 of the reference's 6
    ChebMult + its share of the vector passes) on torch's current stream, timed with HIP events around the K steps; algorithmic bytes per
    launch = 112 P^3 / 3.
    N > 1: a step is pack + the slab launch + exchange + the pencil launch + exchange + the final sum: there is no dominant kernel's launch
    to price, so the object prices the whole STEP of one GPU (its share of the algorithmic bytes and of the flops over the step time) and
    the per-launch fields are null.
    bound: the resource that binds at this size -- FP64 MFMA at P = 256 ((P-2) flop/point per direction against 8-24 B/point), HBM
    at P <= 128.  achieved / peak / frac stay the figure BASELINE.json's metric names ("GB/s vs HBM roofline": SURVEY 8(d) algorithmic bytes
    over time); mfma_f64_frac is the binding one at 256."""
    npts = float(P) ** 3
    step_s = (dev_ms * 1e-3) / steps
    per_launch = world == 1
    unit_s = step_s / launches_per_step if per_launch else step_s
    alg_bytes = BYTES_PER_POINT * npts / (launches_per_step if per_launch else 1) / world
    achieved = alg_bytes / unit_s
    # FP64 MFMA work actually issued: one (P-2)-point even/odd product per direction: 2 (P/2)^2 x 2 halves per line = (P-2) flop/point
    flops = float(P - 2) * (float(P - 2) ** 3) / world * (1.0 if per_launch else 3.0)
    return {"bound": "mfma" if P > 128 else "hbm", "achieved": achieved / 1e9, "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": achieved / HBM_PEAK, "traffic": traffic,
            "frac_is": "model figure of merit: SURVEY 8(d) algorithmic bytes (112 B/point) / time / 8 TB/s, not bytes moved; the binding resource is `bound`: see mfma_f64_frac (FP64 MFMA issued / 78.6 TF) and hbm_real_frac (bytes moved / time / 8 TB/s)",
            "hbm_real_frac": (traffic / unit_s / HBM_PEAK) if traffic else None,
            "kernel": "cheb_sweep_vec4_kernel" if per_launch else
                      "slab route, whole step of one GPU: k_pack, cheb_sweep_multi_kernel (local directions on the slab), cheb_sweep_vec4_kernel (pencil), k_combine, 2 exchanges",
            "priced_per": "launch" if per_launch else "step of one GPU",
            "launches_per_step": launches_per_step, "exchanges_per_step": 0 if per_launch else 2,
            "avg_launch_us": unit_s * 1e6 if per_launch else None, "step_us": step_s * 1e6,
            "algorithmic_bytes_per_launch": alg_bytes if per_launch else None,
            "algorithmic_bytes_per_gpu_step": alg_bytes * (launches_per_step if per_launch else 1),
            "mfma_f64_tflops": flops / unit_s / 1e12, "mfma_f64_frac": flops / unit_s / FP64_MFMA_PEAK}


def csrc_hash():
    """sha256 over the kernel sources (spectral-petsc_amd/csrc/*.hip, *.h, *.cpp, Makefile): ties a committed
    counter measurement (profiles/traffic.json) to the code it was taken on."""
    import glob, hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "spectral-petsc_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h")) + glob.glob(os.path.join(d, "*.cpp")) + [os.path.join(d, "Makefile")]):
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()


def cpu_baseline(P, threads, U, V=None, warm=2, reps=5):
    """Oracle (port of chebyshev.c + MatMult_Elliptic pass structure) on the host, on the bench's own input U, by the
    protocol of BASELINE.md section 3: `warm` untimed applies, `reps` timed ones, median and minimum; buffers are
    recycled between applies, so allocation, first touch and table construction stay outside the timed applies (the
    reference allocates at MatCreate time).  With V (the GPU result for U) the oracle's output is also the full-size
    parity check."""
    import numpy as np
    import oracle_lib as orc
    dims = (P, P, P)
    # genuine FFTW where the box has libfftw3.so.3 (looked up at run time): the reference's own guru plans
    # inside the restated pass structure; otherwise the restated transforms
    use_fftw = threads == 1 and orc.fftw_available()
    t0 = time.perf_counter()
    ref, secs = orc.elliptic_mult_timed(dims, U, mode=orc.FFTW if use_fftw else orc.FAST, nthreads=threads, warm=warm, reps=reps)
    total = time.perf_counter() - t0
    med = float(np.median(secs))
    out = {"value": 1.0 / med, "unit": "matvecs/s", "cores": threads, "kind": "port", "fftw": bool(use_fftw),
           "best": 1.0 / min(secs), "timed_applies": reps, "warmup_applies": warm, "seconds_per_apply": secs,
           "sample": "%d warm-up + %d timed full %d^3 Poisson matvecs (6 ChebMult + pointwise passes each), median%s; oracle %s; %.1f s of CPU work in all"
                     % (warm, reps, P, " (BASELINE.md section 3 asks 2 + 5: cut to 1 + 3 by the ~30-s bound on the CPU sample, 7-8 s per apply on one core)" if (warm, reps) == (1, 3) else "", "pass structure around libfftw3 guru plans (FFTW_ESTIMATE)" if use_fftw else "FAST path (restated transforms; no libfftw3 on this box)", total)}
    parity = None
    if V is not None:
        parity = {"rel_l2_vs_oracle": float(np.linalg.norm(V - ref) / np.linalg.norm(ref)), "tolerance": 1e-10,
                  "input": "the timed %d^3 N(0,1) vector" % P}
    return out, parity


def cpu_baseline_configs(ncpu):
    """BASELINE.md section 3 beside configs 2, 4 and 5 (the headline's 256^3 is `cpu_baseline`): the oracle port on this box's
    host, on seeded N(0,1) inputs of the same shape as the GPU extras -- Poisson 64^3 / 128^3 matvec, StokesMatMult and
    StokesFunction at 64^3 (linear) and 128^3 (power law, -exponent 3 -eps 1e-4).  Protocol 2 warm-up + 5 timed applies,
    median, wherever that fits ~10 s; the legs that do not (one core at 128^3: the 128-point DCT-I is a length-254 = 2 x 127
    (prime) real DFT, 10 s per Poisson apply with the port's generic-radix transform; 128^3 Stokes) run 1 + 2 and say so.
    A port, not FFTW: never a speed-up claim."""
    import numpy as np
    import oracle_lib as orc
    rng = np.random.default_rng(SEED)
    out = {}

    def timed(fn, warm, reps):
        for _ in range(warm):
            fn()
        secs = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); secs.append(time.perf_counter() - t0)
        return {"value": 1.0 / float(np.median(secs)), "unit": "applies/s", "best": 1.0 / min(secs), "warmup_applies": warm, "timed_applies": reps,
                "seconds_per_apply": secs, "kind": "port", "fftw": False}
    for P, legs in ((64, ((1, 2, 5), (ncpu, 2, 5))), (128, ((1, 1, 2), (ncpu, 2, 5)))):
        dims = (P, P, P)
        U = rng.standard_normal((P - 2) ** 3)
        for cores, warm, reps in legs:
            _, secs = orc.elliptic_mult_timed(dims, U, mode=orc.FAST, nthreads=cores, warm=warm, reps=reps)
            out["poisson_%d_matvec_%dcore" % (P, cores)] = {"value": 1.0 / float(np.median(secs)), "unit": "matvecs/s", "best": 1.0 / min(secs), "cores": cores,
                                                              "warmup_applies": warm, "timed_applies": reps, "seconds_per_apply": secs, "kind": "port", "fftw": False}
    power = (1, 1.0, 3.0, 1e-4, 1.0)
    for P, rheo, cores, warm, reps, key in ((64, (0, 1.0, 1.0, 1.0, 1.0), 1, 2, 5, "stokes_64_linear"), (64, (0, 1.0, 1.0, 1.0, 1.0), ncpu, 2, 5, "stokes_64_linear"),
                                            (128, power, ncpu, 1, 2, "stokes_128_powerlaw")):
        dims = (P, P, P)
        N, I, gv, gp, g, dvn = orc.stokes_sizes(dims)
        x = rng.standard_normal(g); dv = np.zeros(dvn); f = np.zeros(g)
        y, eta, deta, strain = orc.stokes_function(dims, x, dv, f, rheology=rheo, mode=orc.FAST, nthreads=max(cores, 4))     # the state of the Jacobian apply
        r = timed(lambda: orc.stokes_function(dims, x, dv, f, rheology=rheo, mode=orc.FAST, nthreads=cores), warm, reps); r["cores"] = cores
        out["%s_function_%dcore" % (key, cores)] = r
        r = timed(lambda: orc.stokes_mult(dims, x, eta=eta, deta=deta, strain=strain, mode=orc.FAST, nthreads=cores), warm, reps); r["cores"] = cores
        out["%s_matmult_%dcore" % (key, cores)] = r
    return out


def _all_ok(ok, dist, torch, backend):
    """True iff every rank reports success (a rank that failed must not leave the others inside a collective)."""
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def dist_parity_check(make, dist, torch, rank, world, backend, P=34, fatal=True):
    """N > 1: the slab-partitioned matvec at a reduced size (P^3) against the oracle's serial matvec, checked on rank 0.
    Returns ({"rel_l2_vs_oracle": .., "P": ..}, None), or (None, reason) when the implementation could not be set up or
    run on some rank (every rank then gets the same answer); a numerical mismatch aborts the bench -- or, with fatal=False
    (a route that has a fall-back), comes back as a reason on every rank."""
    import numpy as np
    op, U, V, why = None, None, None, None
    try:
        op = make((P, P, P))
    except Exception as e:                                  # e.g. RCCL bootstrap of the C-side host refused
        why = "setup: %r" % (e,)
    if not _all_ok(op is not None, dist, torch, backend):
        if op is not None and hasattr(op, "destroy"):
            op.destroy()
        return None, why or "setup failed on another rank"
    try:
        # two calls on the SAME arrays with different contents, the second one checked: a direct route that served a peer's previous
        # values from a cache (the peers rewrite the same addresses call after call) would pass a single call
        U = op.random_input(SEED + 7)
        V = torch.empty_like(U)
        op.mult(U, V)
        U.copy_(op.random_input(SEED + 1))
        op.mult(U, V)
        torch.cuda.synchronize()
    except Exception as e:
        why, V = "matvec: %r" % (e,), None
    if not _all_ok(V is not None, dist, torch, backend):
        return None, why or "matvec failed on another rank"
    pieces = [None] * world
    dist.all_gather_object(pieces, V.cpu().numpy())
    if hasattr(op, "destroy"):
        op.destroy()
    out = None
    if rank == 0:
        import oracle_lib as orc
        g = torch.Generator(device="cpu").manual_seed(SEED + 1)
        Ufull = torch.randn((P - 2) ** 3, dtype=torch.float64, generator=g).numpy()
        ref = orc.elliptic_mult((P, P, P), Ufull, mode=orc.FAST, nthreads=4)
        err = float(np.linalg.norm(np.concatenate(pieces) - ref) / np.linalg.norm(ref))
        out = {"rel_l2_vs_oracle": err, "tolerance": 1e-10, "P": P, "ranks": world}
        if not err <= 1e-10 and fatal:
            raise SystemExit("parity failure: %d-rank %d^3 matvec differs from the oracle by %.3e" % (world, P, err))
    if not fatal:
        box = [None if out is None or out["rel_l2_vs_oracle"] <= 1e-10 else "parity: the %d^3 matvec differs from the oracle by %.3e" % (P, out["rel_l2_vs_oracle"])]
        dist.broadcast_object_list(box, src=0)
        if box[0] is not None:
            return None, box[0]
    return out, None


def extras(sp, torch):
    """Secondary timings (us per call, sustained loops, HIP events) of the other callbacks on the hot path, taken
    AFTER the timed region of the metric: BASELINE configs 2, 4, 5 and the variable-coefficient callbacks at the
    metric's size.  Informational; DESIGN.md section 6 has the byte models they are priced against."""
    import numpy as np

    def t_us(fn, reps):
        # untimed calls first: at least ~20 ms of this very load (the clocks of a GPU that has idled through the CPU baseline take that
        # long to settle: the first timed loop of a callback otherwise reads 3 % high), then the `reps` timed calls as THREE loops, of
        # which the median is reported (one record run of round 6 read FormFunction at 1 068 us where every other run reads 504-535:
        # a single disturbed loop must not stand for the callback)
        t0 = time.perf_counter(); n = 0
        while n < max(reps // 4, 5) or (time.perf_counter() - t0 < 0.02 and n < 2000):
            fn(); n += 1
            if n % 16 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        per = max(reps // 3, 3)
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(per):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / per)
        return sorted(ts)[1]
    out = {}
    rnd = lambda n: torch.randn(n, dtype=torch.float64, device="cuda")
    # (the Stokes handles first: their ~30 work arrays then come from a fresh allocator state -- handles made after gigabytes have been
    # allocated and freed run 5-7 % slower for life)
    for P, power, key in ((64, False, "stokes_64_linear"), (128, True, "stokes_128_powerlaw")):
        op = sp.StokesOp((P, P, P))
        if power:
            op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
        op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
        xs = rnd(op.global_size); ys = torch.empty_like(xs)
        out[key + "_function_us"] = t_us(lambda: op.function(xs, ys), 60)
        out[key + "_matmult_us"] = t_us(lambda: op.mult(xs, ys), 100)
        if not power:
            # config 4 names the Schur apply as well: StokesMatMultSchur = VP, the inner velocity solve on StokesMatMultVV, PV
            # (stokes.C:523-535).  Priced at a FIXED 20 inner GMRES iterations (built-in solver, no preconditioner, restart 30):
            # how many a run takes depends on its inner preconditioner and tolerance, the cost per iteration does not
            xv = rnd(op.velocity_size); yv = torch.empty_like(xv); xp = rnd(op.pressure_size); yp = torch.empty_like(xp)
            out[key + "_matmult_vv_us"] = t_us(lambda: op.mult_vv(xv, yv), 100)
            op.mult_schur(xp, yp, restart=30, rtol=1e-300, max_it=20)
            assert op.inner_iterations == 20
            out[key + "_schur_apply_20_inner_its_us"] = t_us(lambda: op.mult_schur(xp, yp), 10)
        op.destroy()
    op = sp.EllipticOp((128, 128, 128)); U = rnd(op.global_size); V = torch.empty_like(U)
    out["poisson_128_matvec_us"] = t_us(lambda: op.mult(U, V), 300)
    op.destroy()
    op = sp.EllipticOp((256, 256, 256)); U = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5
    X = rnd(op.global_size); b = rnd(op.global_size); R = torch.empty_like(U)
    out["formfunction_256_gamma4_us"] = t_us(lambda: op.function(U, b, R, 4.0, 2.0), 40)
    out["jacobian_apply_256_gamma4_us"] = t_us(lambda: op.mult(X, R), 60)
    op.destroy()
    x = rnd(256 ** 3); y = torch.empty_like(x); pl = sp.ChebPlan((256, 256, 256), 1)
    out["chebmult_256_us"] = t_us(lambda: pl.mult(x, y), 100)
    pl.destroy(); del x, y
    return out


# SURVEY 8(d) byte models (B per grid point) of the callbacks timed by extras(): model us = bytes / 8 TB/s
EXTRAS_MODEL_BYTES = {
    "poisson_128_matvec_us": 112.0 * 128 ** 3, "formfunction_256_gamma4_us": 160.0 * 256 ** 3, "jacobian_apply_256_gamma4_us": 208.0 * 256 ** 3,
    "chebmult_256_us": 16.0 * 256 ** 3, "stokes_64_linear_function_us": 712.0 * 64 ** 3, "stokes_64_linear_matmult_us": 600.0 * 64 ** 3,
    "stokes_128_powerlaw_function_us": 712.0 * 128 ** 3, "stokes_128_powerlaw_matmult_us": 680.0 * 128 ** 3,
}


def extras_frac(ex):
    """Model time (SURVEY 8d algorithmic bytes / 8 TB/s) divided by the measured time of every entry of extras()."""
    return {k.replace("_us", ""): (EXTRAS_MODEL_BYTES[k] / HBM_PEAK * 1e6) / v for k, v in ex.items() if k in EXTRAS_MODEL_BYTES and v}


def dist_rank_compute(sp, dsp, torch, t1_us):
    """The compute side of ONE rank of the slab partition at G = 2, 4, 8, timed alone on this GPU: the handle of rank 0
    (the largest slab) with the NULL transport (chebhip_comm_create_null: every "peer" is the rank itself), so that a call runs the
    launch that fills the pencil (k_pull_pack), the local sweeps, the pencil launch(es) and the final sum (k_pull_combine) of that
    rank -- the kernels of the direct route of csrc/dist.hip -- with every byte read locally and no wire.  T_1 / T_rank is the compute-side
    bound on the speed-up of G ranks; link time is UNMEASURED on hardware (no multi-GPU box) and comes on top.
    Config 3: chebhip_dist_mult on 256^3; config 5: StokesFunction + StokesMatMult on 128^3 power-law slabs."""
    import numpy as np

    def t_us(fn, reps=60):
        # at least ~30 ms of this very load first (a freshly made handle's first loop otherwise reads 5-10 % high: clocks and caches of a
        # 50-us call settle slowly), then three timed loops: the median
        t0 = time.perf_counter(); n = 0
        while n < 15 or (time.perf_counter() - t0 < 0.03 and n < 3000):
            fn(); n += 1
            if n % 32 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / reps)
        return sorted(ts)[1]
    out = {"poisson_256": {"T1_us": t1_us}, "stokes_128_powerlaw": {}, "note": "one rank's kernels alone, no wire: compute-side bound T1 / T_rank; links unmeasured on hardware.  rank_us: every peer's array is the rank's own (the remote rows are cache hits); *_distinct_peer_arrays: G - 1 arrays of their own stand for the peers'"}
    for G in (2, 4, 8):
        comm = dsp.Comm(sp, null=(G, 0))
        D = dsp.DistPoissonC((256, 256, 256), sp, comm=comm)
        U = torch.randn(D.local_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
        # default mode (round 6): the stream policy follows the transport -- nothing leaves the device here, so the local sweeps stay on
        # the caller's stream; rank_us_two_streams (dist_single_stream = 2) is what the side stream costs when there is no wire to hide
        t = t_us(lambda: D.mult(U, V))
        sp.set_option("dist_single_stream", 2)
        try:
            t2s = t_us(lambda: D.mult(U, V))
        finally:
            sp.set_option("dist_single_stream", 0)
        rec = {"rank_us": t, "rank_us_single_stream": t, "rank_us_two_streams": t2s, "bound_speedup": t1_us / t}
        # several vectors per exchange (chebhip_dist_mult_batch): rank time PER VECTOR at nrhs = 2, 4 -- one launch per direction on the
        # stacked slabs / pencils, so the fixed cost of a launch of 256-point lines is shared
        for nrhs in (2, 4):
            Ub = torch.randn((nrhs, D.local_size), dtype=torch.float64, device="cuda"); Vb = torch.empty_like(Ub)
            tb = t_us(lambda: D.mult_batch(Ub, Vb))
            rec["nrhs%d_rank_us_per_vector" % nrhs] = tb / nrhs
            del Ub, Vb
        rec["bound_speedup_nrhs4"] = t1_us / rec["nrhs4_rank_us_per_vector"]
        # ... and with the "peers'" slabs and result arrays as G - 1 arrays of their own (chebhip_comm_null_set_shadow): the pencil job's
        # rows then come from and go to distinct local memory instead of being re-reads of the rank's own planes that hit the caches --
        # still no wire, but no free remote memory either: the tighter of the two compute-side bounds
        try:
            for nrhs in (1, 4):
                Ub = torch.randn((nrhs, D.local_size), dtype=torch.float64, device="cuda"); Vb = torch.empty_like(Ub)
                for k in (0, 1):
                    comm.set_null_shadow(k, [None] + [torch.randn((nrhs, D.local_size), dtype=torch.float64, device="cuda") for _ in range(G - 1)])
                td = t_us((lambda: D.mult(Ub[0], Vb[0])) if nrhs == 1 else (lambda: D.mult_batch(Ub, Vb)))
                rec["rank_us_distinct_peer_arrays" if nrhs == 1 else "nrhs4_rank_us_per_vector_distinct_peer_arrays"] = td / nrhs
                del Ub, Vb
            rec["bound_speedup_distinct_peer_arrays"] = t1_us / rec["rank_us_distinct_peer_arrays"]
            rec["bound_speedup_nrhs4_distinct_peer_arrays"] = t1_us / rec["nrhs4_rank_us_per_vector_distinct_peer_arrays"]
        finally:
            for k in (0, 1):
                comm.set_null_shadow(k, [None] * G)
        out["poisson_256"]["G%d" % G] = rec
        D.destroy(); comm.destroy(); del U, V
    ser = sp.StokesOp((128, 128, 128)); ser.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
    ser.set_dirichlet(np.zeros(ser.dirichlet_size)); ser.set_force(np.zeros(ser.global_size))
    xs = torch.randn(ser.global_size, dtype=torch.float64, device="cuda"); ys = torch.empty_like(xs)
    tf1, tm1 = t_us(lambda: ser.function(xs, ys)), t_us(lambda: ser.mult(xs, ys))
    ser.destroy(); del xs, ys
    out["stokes_128_powerlaw"]["T1_us"] = {"function": tf1, "matmult": tm1}
    for G in (2, 4, 8):
        comm = dsp.Comm(sp, null=(G, 0))
        D = dsp.DistStokesC((128, 128, 128), sp, comm=comm)
        D.op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
        D.op.set_dirichlet(np.zeros(D.dirichlet_size)); D.op.set_force(np.zeros(D.global_size))
        x = torch.randn(D.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
        tf, tm = t_us(lambda: D.function(x, y)), t_us(lambda: D.mult(x, y))
        out["stokes_128_powerlaw"]["G%d" % G] = {"function_rank_us": tf, "matmult_rank_us": tm, "bound_speedup_function": tf1 / tf, "bound_speedup_matmult": tm1 / tm}
        D.destroy(); comm.destroy(); del x, y
    return out


def package_power(step, torch, seconds=2.5):
    """Package power (W) of GPU 0 as rocm-smi reports it, sampled while the metric's matvec loops for a few seconds, and
    once more after a pause: the headline kernel runs into the board's power cap, so the figure belongs
    next to the rate.  Informational; None when rocm-smi is not usable."""
    import re
    import subprocess
    import threading

    def read():
        try:
            out = subprocess.run(["rocm-smi", "--showpower"], capture_output=True, text=True, timeout=20).stdout
            m = re.search(r"GPU\[0\].*?Power \(W\):\s*([0-9.]+)", out)
            return float(m.group(1)) if m else None
        except Exception:
            return None
    vals = []
    stop = [False]

    def sampler():
        time.sleep(0.8)
        while not stop[0]:
            v = read()
            if v is not None:
                vals.append(v)
            time.sleep(0.3)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            step()
        torch.cuda.synchronize()
    stop[0] = True
    th.join()
    time.sleep(1.5)
    idle = read()
    if not vals:
        return None
    return {"package_w_under_matvec_loop": max(vals), "package_w_after_pause": idle, "samples": len(vals), "source": "rocm-smi --showpower, GPU[0]"}


def solves(sp, torch):
    """End-to-end solves of the callers of the hot path at the BASELINE sizes, after the timed region:
    Newton + FGMRES(30) + the finite-difference preconditioner on the 256^3 elliptic problem (-gamma 4 -exponent 2),
    the linear Stokes solve at 64^3 (config 4) and the power-law Stokes solve with continuation at 128^3 (config 5,
    README:52) with the block preconditioner StokesPCApply0.  Manufactured: a smooth field x* with zero boundary values is
    put through the operator itself to get the right-hand side, the solve starts from 0 and must come back to x*.
    Wall seconds (synchronised) and the error against x*; informational."""
    import importlib
    import numpy as np
    solve = importlib.import_module(sp.__name__ + ".solve")
    out = {}

    def nodes(P):
        return np.cos(np.pi * np.arange(1, P - 1) / (P - 1))

    def smooth(P, d, seed):
        x = nodes(P)
        rng = np.random.default_rng(seed)
        f = np.ones((P - 2,) * d)
        for k in range(d):
            a, b = rng.random(2) + 0.5
            g = (1.0 - x * x) * (1.0 + 0.3 * np.cos(2.0 * a * x + b))
            f = f * g.reshape([-1 if j == k else 1 for j in range(d)])
        return f

    # --- elliptic 256^3, gamma 4 (tests.sh:10 parameters)
    P = 256
    op = sp.EllipticOp((P, P, P))
    op.set_dirichlet(np.zeros(op.dirichlet_size))
    us = torch.from_numpy(smooth(P, 3, 1).ravel()).cuda()
    b = torch.empty_like(us)
    op.function(us, None, b, 4.0, 2.0)                      # b = A(u*) u*
    x = torch.zeros_like(us)
    # Every solve is run twice from the same start and the SECOND run is reported (`seconds`): the first (`seconds_first_run`)
    # also pays the one-off costs of a fresh process -- device allocations of the Krylov bases (gigabytes at these sizes), first
    # touches, clocks -- which vary by tens of per cent from box to box and say nothing about the solve (1.28 / 1.51 s were seen
    # for the same config-5 solve on the same sources).  Same iteration counts in both runs.
    # The Krylov handle (8 GB of basis vectors at this size) is the caller's and outlives the solves, as the KSP of elliptic.C:181-185
    # does: making and freeing it inside the timed region added 0.0-0.8 s of hipMalloc / hipFree at random (device time of the solve
    # constant at 0.22 s: tools/solve_order_probe.py, profiles/r05_solve_order.txt).
    dts = []
    ks = sp.Fgmres(op.global_size, restart=30, rtol=1e-6, max_it=300)
    for rep in range(2):
        x.zero_()
        pc = sp.FdPc(op, sweeps=0)                          # (a fresh preconditioner per run: assembled from the state of ITS first residual)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        its, kits, fn = solve.newton_krylov(sp, op, b, x, 4.0, 2.0, snes_rtol=1e-10, ksp_rtol=1e-6, ksp_restart=30, ksp_max_it=300, M=pc,
                                            monitor=lambda i, f, k: pc.update(), ks=ks)
        torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
        if rep == 0:
            pc.destroy()
    ks.destroy()
    out["elliptic_256_gamma4"] = {"seconds": dts[1], "seconds_first_run": dts[0], "newton_its": its, "krylov_its": kits,
                                  "rel_err_vs_manufactured": float((x - us).abs().max() / us.abs().max())}
    pc.destroy(); op.destroy(); del us, b, x

    # --- Stokes: linear 64^3 (config 4), power law with continuation 128^3 (config 5), on the reference's own manufactured
    # problem -exact 2 (StokesExact2, stokes.C:1963-2012; README:43,52): u = sin(pi x/2) cos(pi y/2), v = -cos(pi x/2) sin(pi y/2),
    # w = 0, p = 0, force (pi/2)^2 (u, v, 0, 0), Dirichlet values = the field on the boundary nodes (row-major node order,
    # components innermost).  For the power law the force stays the linear one, as in the reference's run.
    def exact2(P):
        c = np.cos(np.pi * np.arange(P) / (P - 1))
        X, Y, Z = np.meshgrid(c, c, c, indexing="ij")
        u = np.sin(0.5 * np.pi * X) * np.cos(0.5 * np.pi * Y); v = -np.cos(0.5 * np.pi * X) * np.sin(0.5 * np.pi * Y)
        val = np.stack([u, v, np.zeros_like(u), np.zeros_like(u)], axis=-1).reshape(-1, 4)
        idx = np.arange(P)
        bd1 = (idx == 0) | (idx == P - 1)
        bd = (bd1[:, None, None] | bd1[None, :, None] | bd1[None, None, :]).ravel()
        U = val[~bd]
        rhs = U.copy(); rhs[:, :2] *= (0.5 * np.pi) ** 2; rhs[:, 2:] = 0.0
        return U.ravel(), rhs.ravel(), np.ascontiguousarray(val[bd][:, :3]).ravel()

    for P, rheo, cont, key in ((64, (0, 1.0, 1.0, 1.0, 1.0), 1, "stokes_64_linear_exact2"), (128, (1, 1.0, 3.0, 1e-4, 1.0), 4, "stokes_128_powerlaw_cont4_exact2")):
        st = sp.StokesOp((P, P, P))
        U, U2, dv = exact2(P)
        st.set_dirichlet(dv); st.set_force(U2)
        x = torch.zeros(st.global_size, dtype=torch.float64, device="cuda")
        dts = []
        ks = sp.Fgmres(st.global_size, restart=60, rtol=1e-5 if rheo[0] else 1e-12, max_it=200)       # the caller's KSP and PC objects (stokes.C:155-176)
        pcs = sp.StokesSaddlePc(st, 0)
        for rep in range(2):                                # (the second run is the one reported: see above)
            x.zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            stats = {}
            # the linear problem is solved to its floor (-exact 2 is resolved to rounding on 64 CGL points): tight tolerances
            log = solve.stokes_solve(sp, st, x, rheology=rheo, cont0=0, cont=cont, snes_rtol=1e-8 if rheo[0] else 1e-12, ksp_rtol=1e-5 if rheo[0] else 1e-12,
                                     ksp_restart=60, ksp_max_it=200, max_linear_fail=3, snes_max_it=20, stats=stats, ks=ks, pc=pcs)
            torch.cuda.synchronize(); dts.append(time.perf_counter() - t0)
        ks.destroy(); pcs.destroy()
        rec = {"seconds": dts[1], "seconds_first_run": dts[0], "stages": len(log), "newton_its": int(sum(s[2] for s in log)), "krylov_its": int(sum(s[3] for s in log)),
               "residual_norm": float(log[-1][4]), "linear_solves_ended_on_iteration_limit": int(stats.get("linear_fails", -1))}
        if not rheo[0]:                                     # the field is the exact solution of the linear problem only
            xs = x.cpu().numpy().reshape(-1, 4); Us = U.reshape(-1, 4)
            rec["max_velocity_err_vs_exact2"] = float(np.abs(xs[:, :3] - Us[:, :3]).max())
            rec["err_is"] = "solver floor: the field is resolved to rounding on this grid, the error is what snes/ksp tolerances and the conditioning leave"
        out[key] = rec
        st.destroy(); del x
    return out


def local_threads_main(args):
    """BENCH_DIST_TRANSPORT=local python bench.py --gpus N: ONE process, N rank THREADS, thread r on device r % (devices on the node) with its
    own stream, the LOCAL transport of csrc/comm.hip (peer access between the devices; no RCCL, no messages: a rank's kernels read the
    peers' slabs and pencil results in place, two thread rendezvous per matvec -- csrc/dist.hip "direct").  The same W / K protocol as the
    process-per-GPU route: thread barrier + device synchronisation on both sides of the K timed steps, the MAX over the ranks, one line.
    On a one-GPU box every rank lands on device 0: plumbing and parity, not a measurement (labelled in config.parallelism).  Never a re-exec:
    the threads are started by this process before anything else has touched a GPU."""
    import threading
    import numpy as np
    import torch
    import __graft_entry__ as ge
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    sp = ge.load(); dsp = ge.load_dist()
    G, P = args.gpus, args.size
    ndev = torch.cuda.device_count()
    lg = dsp.LocalGroup(sp, G)
    bar = threading.Barrier(G)
    walls, devms, errs, pieces, box = [0.0] * G, [0.0] * G, [None] * G, [None] * G, {}

    def sync_all():
        torch.cuda.synchronize(); bar.wait(); torch.cuda.synchronize()

    def worker(r):
        comm = None
        try:
            torch.cuda.set_device(r % ndev)
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                comm = lg.comm(r)
                # reduced-size parity against the oracle (rank 0 checks the concatenated slabs)
                Ps = 34
                small = dsp.DistPoissonC((Ps, Ps, Ps), sp, comm=comm)
                Us = small.random_input(SEED + 1); Vs = torch.empty_like(Us)
                small.mult(Us, Vs); st.synchronize()
                pieces[r] = (small.slab_offset, Vs.cpu().numpy())
                small.destroy()
                bar.wait()
                if r == 0:
                    import oracle_lib as orc
                    g = torch.Generator(device="cpu").manual_seed(SEED + 1)
                    Uf = torch.randn((Ps - 2) ** 3, dtype=torch.float64, generator=g).numpy()
                    ref = orc.elliptic_mult((Ps, Ps, Ps), Uf, mode=orc.FAST, nthreads=4)
                    got = np.concatenate([p[1] for p in sorted(pieces, key=lambda t: t[0])])
                    box["parity"] = {"rel_l2_vs_oracle": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)), "tolerance": 1e-10, "P": Ps, "ranks": G}
                op = dsp.DistPoissonC((P, P, P), sp, comm=comm)
                if r == 0:
                    box["local_size"] = op.local_size
                U = op.random_input(SEED); V = torch.empty_like(U)
                for _ in range(args.spinup):
                    op.mult(U, V)
                sync_all()
                for _ in range(args.warmup):
                    op.mult(U, V)
                sync_all()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(st)
                for _ in range(args.steps):
                    op.mult(U, V)
                e1.record(st)
                sync_all()
                walls[r] = time.perf_counter() - t0
                devms[r] = e0.elapsed_time(e1)
                assert torch.isfinite(V).all()
                op.destroy()
        except BaseException as e:                          # noqa: a failing rank must not leave the others at a barrier
            errs[r] = e
            lg.abort(); bar.abort()
        finally:
            if comm is not None and errs[r] is None:
                comm.destroy()

    wd = Watchdog(0)
    wd.enter("the %d thread ranks of the LOCAL transport" % G, int(os.environ.get("BENCH_WATCHDOG_S", "600")))
    th = [threading.Thread(target=worker, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    wd.leave()
    lg.destroy()
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    for e in errs:
        if e is not None:
            raise e
    par = box["parity"]
    if not par["rel_l2_vs_oracle"] <= par["tolerance"]:
        raise SystemExit("parity failure: %d-rank %d^3 matvec differs from the oracle by %.3e" % (G, par["P"], par["rel_l2_vs_oracle"]))
    wall = max(walls)
    # direct route (csrc/dist.hip): up to 12 M values per rank the three directions are ONE launch (two local jobs + the pencil job reading the
    # peers' slabs in place), then the final sum reading the peers' pencil results: two kernels; above: two local launches + the pencil launch + the sum
    launches = 2 if box["local_size"] < 12000000 else 4
    out = {
        "metric": "spectral matvecs/s and GB/s vs HBM roofline, 3D P^3 grid",
        "value": args.steps / wall, "unit": "matvecs/s", "n_gpus": G, "steps": args.steps, "warmup": args.warmup, "spinup": args.spinup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "3-D Poisson MatMult_Elliptic -dim %d,%d,%d (gamma=0), global N(0,1) input seed %d" % (P, P, P, SEED), "P": P,
                   "parallelism": "slab%d+direct-pull(LOCAL transport: %d thread ranks of one process on %d device(s)%s)" % (
                       G, G, min(G, ndev), "" if ndev >= G else "; ranks SHARE devices: plumbing and parity, not a measurement"),
                   "launches_per_step": launches, "devices": min(G, ndev)},
        "roofline": roofline_record(P, G, args.steps, max(devms), launches, None),
        "device_ms_per_step": max(devms) / args.steps,
        "parity": par,
    }
    out["roofline"]["kernel"] = ("slab route (direct transport), whole step of one GPU: " +
                                 ("cheb_sweep_multi_gather_kernel (ONE launch: the local directions + the pencil direction reading the peers' slabs in place)" if launches == 2 else
                                  "the local directions, cheb_sweep_vec4_gather_kernel (the pencil direction reading the peers' slabs in place)") +
                                 " storing its rows into the owners' result arrays, k_pull_combine (the final sum, local reads); 2 rendezvous, no messages")
    out["roofline"]["exchanges_per_step"] = 0
    out["roofline"]["rendezvous_per_step"] = 2
    print(json.dumps(out), flush=True)
    return 0


def main():
    args = parse()
    if args.gpus > 1 and os.environ.get("BENCH_DIST_TRANSPORT", "") == "local" and not args.launch_selftest:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
            raise SystemExit("BENCH_DIST_TRANSPORT=local drives the N GPUs from ONE process (thread ranks): run `python bench.py --gpus N` "
                             "without a launcher (WORLD_SIZE=%s is set)" % os.environ["WORLD_SIZE"])
        sys.exit(local_threads_main(args))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # typed without a launcher: this process becomes the parent of the N ranks -- decided before torch is imported
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run, or unset WORLD_SIZE and let bench.py start its ranks)" % (args.gpus, world))
    if args.launch_selftest:
        sys.exit(launch_selftest(args, world, rank))
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    sp = ge.load()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if world > 1 and os.environ.get("BENCH_DIST_BACKEND", "nccl") == "nccl" and torch.cuda.device_count() < world:
        raise SystemExit("--gpus %d on a box with %d GPU(s): RCCL needs one GPU per rank (BENCH_DIST_BACKEND=gloo rehearses the N > 1 path "
                         "with the ranks sharing a GPU and the exchange staged through the host: plumbing, not a measurement)" % (world, torch.cuda.device_count()))
    # one rank per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N > 1
    # path on a single-GPU box (the exchange is then staged through the host: not a measurement)
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)

    P = args.size
    dims = (P, P, P)
    wd = Watchdog(rank)
    wd_s = int(os.environ.get("BENCH_WATCHDOG_S", "600"))
    if world > 1:
        wd.enter("transport setup and the reduced-size parity check of the %d-rank matvec" % world, wd_s)
    if world == 1:
        op = sp.EllipticOp(dims)
        g = torch.Generator(device="cuda").manual_seed(SEED)
        U = torch.randn(op.global_size, dtype=torch.float64, device="cuda", generator=g)
        V = torch.empty_like(U)
        step = lambda: op.mult(U, V)
        launches_per_step = 3
        parallelism = "single"
    else:
        dsp = ge.load_dist()
        # The slab driver behind the C ABI (csrc/dist.hip: chebhip_dist_*, RCCL grouped send/recv on its own
        # communicator); BENCH_DIST_IMPL=python selects its Python twin (torch.distributed all_to_all_single).
        impl = os.environ.get("BENCH_DIST_IMPL", "c")
        make_c = lambda dm: dsp.DistPoissonC(dm, sp)
        make_py = lambda dm: dsp.DistPoissonOp(dm, backend=dsp.HipBackend(sp))
        make = make_c if impl == "c" else make_py
        # BENCH_DIST_TRANSPORT (launcher runs, C host): "auto" (default) tries the DIRECT route among the processes of the node
        # (csrc/comm.hip IPC transport over the group's own: the ranks' kernels read each other's slabs in place, no pack, no
        # messages) and keeps it only if the node grants the shared mappings AND the matvec equals the oracle's at two reduced sizes
        # (34^3: pull route; 130^3: gather loader and the one-launch form, the kernels of the full size) -- otherwise the message
        # route (RCCL grouped send / recv), said in config.ipc_fallback.  "messages": the message route; "ipc": the direct route or fail.
        want_tr = os.environ.get("BENCH_DIST_TRANSPORT", "auto")
        if want_tr not in ("auto", "ipc", "messages"):
            raise SystemExit("BENCH_DIST_TRANSPORT must be auto, ipc, messages (launcher runs) or local (one process, thread ranks)")
        ipc_note = None
        dist_parity, why = None, None
        used_ipc = False
        if impl == "c" and want_tr in ("auto", "ipc"):
            # the ranks of a bench run arrive at every rendezvous within milliseconds: a peer that has not come after 30 s will not come
            # (this bounds what a node on which the direct route does not work costs before the fall-back: one time limit, not the default 120 s)
            sp.set_option("local_timeout_s", int(os.environ.get("BENCH_IPC_TIMEOUT_S", "30")))

            def make_ipc(dm):
                o = dsp.DistPoissonC(dm, sp, ipc=True)
                if not o.transport.startswith("ipc"):
                    msg = o._own_comm.ipc_error if o._own_comm is not None else "no communicator"
                    o.destroy()
                    raise RuntimeError("the node did not grant the IPC group: %s" % msg)
                return o
            par_small, why_ipc = dist_parity_check(make_ipc, dist, torch, rank, world, backend, fatal=False)
            if why_ipc is None:
                dist_parity, why_ipc = dist_parity_check(make_ipc, dist, torch, rank, world, backend, P=130, fatal=False)
            if why_ipc is None:
                make, used_ipc = make_ipc, True
                if rank == 0:
                    dist_parity["also"] = par_small
            elif want_tr == "ipc":
                raise SystemExit("BENCH_DIST_TRANSPORT=ipc: %s" % why_ipc)
            else:
                ipc_note = why_ipc
                if rank == 0:
                    print("bench.py: the direct (IPC) route is not used (%s); timing the message route" % why_ipc, file=sys.stderr, flush=True)
        if not used_ipc:
            dist_parity, why = dist_parity_check(make, dist, torch, rank, world, backend)   # reduced size, against the oracle on rank 0
        dist_fallback = None
        if dist_parity is None and why is not None and impl == "c" and os.environ.get("BENCH_DIST_STRICT", "0") != "1":
            # The C-side host could not run on this node (its RCCL peer path has never met more than one rank on hardware: no
            # multi-GPU box exists for the builder).  A scaling record of the same algorithm on its Python host
            # (torch.distributed all_to_all_single) is worth more than none: say so loudly and in the line; BENCH_DIST_STRICT=1 fails instead.
            if rank == 0:
                print("bench.py: the C-side slab host failed (%s); timing its Python twin instead -- see config.c_host_fallback" % why, file=sys.stderr, flush=True)
            dist_fallback = why
            impl, make = "python", make_py
            dist_parity, why = dist_parity_check(make, dist, torch, rank, world, backend)
        if why is not None:
            raise SystemExit("the %d-rank matvec could not be run: %s" % (world, why))
        op = make(dims)
        U = op.random_input(SEED)
        V = torch.empty_like(U)
        step = lambda: op.mult(U, V)
        # pack, the local directions (ONE launch of d - 1 jobs on a slab of fewer than 6 M values, csrc/dist.hip), the pencil launch, the final sum
        launches_per_step = 4 if getattr(op, "local_size", 0) < 6000000 else 5
        parallelism = "slab%d+all2all(%s, %s)" % (world, "C host" if impl == "c" else "python host", backend)
        if used_ipc:
            # the three directions in one launch + the final sum (csrc/dist.hip FUSE3_MAX), else local launch(es), pencil launch, sum;
            # each of the two rendezvous adds one 64-thread polling launch, not counted
            launches_per_step = 2 if op.local_size < 12000000 else 4
            parallelism = "slab%d+direct-pull(IPC transport: one process per GPU, peers' slabs read in place; reductions and segment exchanges over %s)" % (world, backend)
        wd.enter("spin-up, warm-up and the %d timed steps of the %d-rank matvec" % (args.steps, world), wd_s)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # Setup, untimed and outside the W/K protocol: a fixed spin-up so that clocks, caches and the
    # launch path are in steady state before the W warm-up steps (an idle MI355X needs a few ms of
    # load to settle; with W = 3 the first timed steps would otherwise run ~15 % slow).
    for _ in range(args.spinup):
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    t = torch.tensor([wall], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall = t.item()
    wd.leave()
    assert torch.isfinite(V).all()
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = args.steps / wall
        npts = float(P) ** 3
        # HBM bytes per launch from the committed PMC profile -- only if it was taken on the very kernel sources
        # being timed (hash of csrc/), otherwise null
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            if (tj.get("P") == P and world == 1 and tj.get("launches_per_matvec") == launches_per_step
                    and tj.get("csrc_sha256") == csrc_hash()):
                traffic = tj["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "spectral matvecs/s and GB/s vs HBM roofline, 3D P^3 grid",
            "value": value, "unit": "matvecs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup": args.spinup,
            "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3-D Poisson MatMult_Elliptic -dim %d,%d,%d (gamma=0), global N(0,1) input seed %d" % (P, P, P, SEED),
                       "P": P, "parallelism": parallelism, "launches_per_step": launches_per_step},
            "roofline": roofline_record(P, world, args.steps, dev_ms, launches_per_step, traffic),
            "device_ms_per_step": dev_ms / args.steps,
        }
        if world > 1:
            out["parity"] = dist_parity
            if used_ipc:
                out["roofline"]["kernel"] = ("slab route (direct transport among processes), whole step of one GPU: " +
                                             ("cheb_sweep_multi_gather_kernel (ONE launch: the local directions + the pencil direction reading the peers' slabs in place)" if launches_per_step == 2 else
                                              "the local directions, cheb_sweep_vec4_gather_kernel (the pencil direction reading the peers' slabs in place)") +
                                             " storing its rows into the owners' result arrays, k_pull_combine (the final sum, local reads); 2 rendezvous (one 8 x 64-thread polling launch each), no messages")
                out["roofline"]["exchanges_per_step"] = 0
            if dist_fallback:
                out["config"]["c_host_fallback"] = dist_fallback
            if ipc_note:
                out["config"]["ipc_fallback"] = ipc_note[:300]
        if world == 1 and not args.no_cpu_baseline:
            try:
                Uh, Vh = U.cpu().numpy(), V.cpu().numpy()
                # the faithful one (the reference is serial); bounded to ~30 s of CPU work: 1 warm-up + 3 timed applies at one
                # core (7-8 s per 256^3 apply: BASELINE.md section 3's 2 + 5 would be a minute); 2 + 5 on all cores
                one = args.cpu_threads == 1 and P >= 256
                out["cpu_baseline"], out["parity"] = cpu_baseline(P, args.cpu_threads, Uh, Vh, warm=1 if one else 2, reps=3 if one else 5)
                ncpu = min(os.cpu_count() or 1, 16)
                if ncpu > args.cpu_threads:                                       # the generous one: OpenMP over lines
                    out["cpu_baseline_all_cores"], _ = cpu_baseline(P, ncpu, Uh)
                if not args.no_extras:                                            # configs 2, 4, 5 on the host as well
                    out["cpu_baseline_configs"] = cpu_baseline_configs(ncpu)
            except Exception as e:                                                # the checker must not cost the metric line
                out["cpu_baseline"] = {"value": None, "unit": "matvecs/s", "cores": args.cpu_threads, "kind": "port", "sample": "failed: " + repr(e)[:160]}
            if out.get("parity") and out["parity"]["rel_l2_vs_oracle"] > out["parity"]["tolerance"]:
                raise SystemExit("parity failure: GPU matvec differs from the oracle by %.3e" % out["parity"]["rel_l2_vs_oracle"])
        if world == 1 and not args.no_extras:
            try:
                pw = package_power(step, torch)
                if pw:
                    out["power"] = pw
            except Exception as e:
                out["power_error"] = repr(e)[:200]
        if world == 1 and not args.no_extras:
            try:                                            # informational: never at the expense of the metric line
                out["extras_us"] = extras(sp, torch)        # (before solves(): the callbacks' handles are made in a fresh allocator state)
                out["extras_frac"] = extras_frac(out["extras_us"])
            except Exception as e:
                out["extras_us"] = {"error": repr(e)[:200]}
        if world == 1 and not args.no_extras:
            try:
                out["solves"] = solves(sp, torch)
            except Exception as e:
                out["solves_error"] = repr(e)[:300]
        if world == 1 and not args.no_extras:
            try:                                            # compute side of one rank at G = 2, 4, 8 (no wire)
                out["dist_rank_compute"] = dist_rank_compute(sp, ge.load_dist(), torch, dev_ms / args.steps * 1e3)
            except Exception as e:
                out["dist_rank_compute"] = {"error": repr(e)[:200]}
        print(json.dumps(out), flush=True)
    if world > 1:
        # After the line has left (nothing below can cost it), opt-in (BENCH_BATCHED=1): the same matvec on 4 vectors per exchange
        # (chebhip_dist_mult_batch), reported on stderr.  The ranks agree on an error flag after the first batched call, so a rank that
        # failed does not leave its peers in a collective; a leg -- or a final barrier -- that does not return ends under the watchdog
        # with a diagnostic and a NON-ZERO status.
        batched = None
        if hasattr(op, "mult_batch") and os.environ.get("BENCH_BATCHED", "0") == "1":
            wd.enter("the informational batched leg (BENCH_BATCHED=1: 4 vectors per exchange)", min(wd_s, 180))
            nrhs = 4
            err = None
            try:
                Ub = torch.stack([U] * nrhs).contiguous(); Vb = torch.empty_like(Ub)
                op.mult_batch(Ub, Vb)
                torch.cuda.synchronize()
            except Exception as e:
                err = repr(e)[:200]
            if not _all_ok(err is None, dist, torch, backend):
                batched = {"error": err or "failed on another rank"}
            else:
                kb = max(args.steps // nrhs, 2)
                for _ in range(max(args.warmup // nrhs, 2)):
                    op.mult_batch(Ub, Vb)
                barrier()
                tb0 = time.perf_counter()
                for _ in range(kb):
                    op.mult_batch(Ub, Vb)
                barrier()
                tb = torch.tensor([time.perf_counter() - tb0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
                dist.all_reduce(tb, op=dist.ReduceOp.MAX)
                batched = {"nrhs": nrhs, "steps": kb, "matvecs_per_s": kb * nrhs / tb.item(), "ms_per_batch": tb.item() * 1e3 / kb,
                           "vectors_agree": bool(torch.equal(Vb[0], Vb[nrhs - 1]))}
            if rank == 0:
                print("bench.py (informational, not part of the line): %d vectors per exchange: %s" % (nrhs, json.dumps(batched)), file=sys.stderr, flush=True)
        wd.enter("the final barrier", 120)
        dist.barrier()
        dist.destroy_process_group()
        wd.leave()


if __name__ == "__main__":
    main()
