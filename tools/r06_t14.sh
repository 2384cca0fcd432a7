set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_dist_emul.py -x -q -m gpu > gpurun_out/r06_t14_tests.log 2>&1 || { tail -40 gpurun_out/r06_t14_tests.log; exit 1; }
tail -3 gpurun_out/r06_t14_tests.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_t14.log
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
def t_us(fn, reps=60):
    t0 = time.perf_counter(); n = 0
    while n < 15 or time.perf_counter() - t0 < 0.03:
        fn(); n += 1
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    return sorted(ts)[1]
for packed in (1, 0):
    sp.set_option("dist_packed_exchange", packed)
    for G in (8, 4, 2):
        comm = dsp.Comm(sp, null=(G, 0))
        D = dsp.DistStokesC((128, 128, 128), sp, comm=comm)
        D.op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
        D.op.set_dirichlet(np.zeros(D.dirichlet_size)); D.op.set_force(np.zeros(D.global_size))
        x = torch.randn(D.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
        print("packed=%d G=%d: StokesFunction %.1f us  StokesMatMult %.1f us" % (packed, G, t_us(lambda: D.function(x, y)), t_us(lambda: D.mult(x, y))))
        D.destroy(); comm.destroy()
sp.set_option("dist_packed_exchange", 0)
PY
