set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_t2_parity.log 2>&1 || { tail -30 gpurun_out/r06_t2_parity.log; exit 1; }
tail -3 gpurun_out/r06_t2_parity.log
{
for rep in 1 2; do
  for lib in tools/libchebhip_ov0.so tools/libchebhip_fa2.so tools/libchebhip_fa4.so spectral-petsc_amd/libchebhip.so tools/libchebhip_fa8.so tools/libchebhip_fa12.so; do
    echo "== $lib"; CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 256
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 128
  done
done
for lib in tools/libchebhip_ov0.so tools/libchebhip_fa2.so tools/libchebhip_fa4.so spectral-petsc_amd/libchebhip.so tools/libchebhip_fa8.so tools/libchebhip_fa12.so; do
  echo "== $lib"
  CHEBHIP_LIB_PATH=$lib timeout -k 10 120 python tools/dist_rank_trace.py 8 200 dist_single_stream=1
done
for v in diag_ov0 diag_fa2 diag_fa4 diag diag_fa8 diag_fa12; do
timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_$v.so
done
timeout -k 10 120 python tools/stamp_probe3.py 128 tools/libchebhip_diag.so
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t2_ab.log
grep -B1 "P=\|G = " gpurun_out/r06_t2_ab.log
