set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests/test_gpu_dist_emul.py -x -q -m gpu -k "thread_ranks_small or direct_pull or batch or poisson_256" > gpurun_out/r06_t11_tests.log 2>&1 || { tail -40 gpurun_out/r06_t11_tests.log; exit 1; }
tail -3 gpurun_out/r06_t11_tests.log
{
for G in 8 4 2; do
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200
  timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_packed_exchange=2
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t11.log
cat gpurun_out/r06_t11.log
