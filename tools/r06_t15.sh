set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_dist_emul.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r06_t15_tests.log 2>&1 || { tail -40 gpurun_out/r06_t15_tests.log; exit 1; }
tail -3 gpurun_out/r06_t15_tests.log
