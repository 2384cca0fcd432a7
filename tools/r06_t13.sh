set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
for G in 8 4 2; do
  timeout -k 10 200 python tools/dist_rank_batch.py $G 1 2 4
  timeout -k 10 200 python tools/dist_rank_batch.py $G 1 2 4 dist_packed_exchange=3
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t13.log
cat gpurun_out/r06_t13.log
timeout -k 10 600 python -m pytest tests/test_gpu_dist_emul.py -x -q -m gpu -k "poisson" > gpurun_out/r06_t13_tests.log 2>&1 || { tail -40 gpurun_out/r06_t13_tests.log; exit 1; }
tail -3 gpurun_out/r06_t13_tests.log
