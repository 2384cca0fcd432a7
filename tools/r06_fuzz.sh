set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python -u tools/fuzz_parity.py 360 606 > gpurun_out/r06_fuzz_parity.log 2>&1 || { tail -30 gpurun_out/r06_fuzz_parity.log; exit 1; }
tail -4 gpurun_out/r06_fuzz_parity.log
timeout -k 10 500 python -u tools/fuzz_dist_threads.py 300 606 > gpurun_out/r06_fuzz_dist_threads.log 2>&1 || { tail -30 gpurun_out/r06_fuzz_dist_threads.log; exit 1; }
tail -4 gpurun_out/r06_fuzz_dist_threads.log
