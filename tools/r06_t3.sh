set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
bash tools/r06_lds_probe.sh gpurun_out/r06_lds_probe
timeout -k 10 120 python tools/stamp_probe_multi.py 64 tools/libchebhip_diag.so
timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag.so
timeout -k 10 120 python tools/stokes_bench.py
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t3.log
cat gpurun_out/r06_t3.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r06_t3_gputest.log 2>&1 || { tail -40 gpurun_out/r06_t3_gputest.log; exit 1; }
tail -3 gpurun_out/r06_t3_gputest.log
