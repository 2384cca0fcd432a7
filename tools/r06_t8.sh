set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 1100 python -m pytest tests/test_gpu_dist_emul.py tests/test_gpu_dist.py tests/test_bench_launch.py tests/test_gpu_aux.py -x -q -m gpu > gpurun_out/r06_t8_tests.log 2>&1 || { tail -40 gpurun_out/r06_t8_tests.log; exit 1; }
tail -3 gpurun_out/r06_t8_tests.log
{
for G in 8 4 2; do
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_packed_exchange=1
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_single_stream=2
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t8.log
cat gpurun_out/r06_t8.log
