set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 600 python -u tools/fuzz_dist.py 300 606 > gpurun_out/r06_fuzz_dist_procs.log 2>&1 || { tail -30 gpurun_out/r06_fuzz_dist_procs.log; exit 1; }
tail -6 gpurun_out/r06_fuzz_dist_procs.log
