set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_dist_emul.py tests/test_gpu_aux.py tests/test_gpu_stokes.py -x -q -m gpu > gpurun_out/r06_t12_tests.log 2>&1 || { tail -40 gpurun_out/r06_t12_tests.log; exit 1; }
tail -3 gpurun_out/r06_t12_tests.log
{
for G in 8 4 2; do
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_packed_exchange=3
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_packed_exchange=2
done
timeout -k 10 120 python tools/stokes_bench.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t12.log
cat gpurun_out/r06_t12.log
