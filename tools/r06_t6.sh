set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_aux.py tests/test_gpu_stokes.py -x -q -m gpu > gpurun_out/r06_t6_parity.log 2>&1 || { tail -30 gpurun_out/r06_t6_parity.log; exit 1; }
tail -3 gpurun_out/r06_t6_parity.log
{
for rep in 1 2 3; do
  timeout -k 10 120 python tools/quick_bench.py 256 poisson_launches=2
  timeout -k 10 120 python tools/quick_bench.py 256
done
timeout -k 10 120 python tools/quick_bench.py 200 poisson_launches=2
timeout -k 10 120 python tools/quick_bench.py 200
timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag.so
timeout -k 10 120 python tools/stokes_bench.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t6.log
cat gpurun_out/r06_t6.log
