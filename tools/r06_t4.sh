set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stokes.py -x -q -m gpu > gpurun_out/r06_t4_parity.log 2>&1 || { tail -30 gpurun_out/r06_t4_parity.log; exit 1; }
tail -3 gpurun_out/r06_t4_parity.log
{
bash tools/r06_lds_probe.sh gpurun_out/r06_lds_probe
for rep in 1 2; do
  for lib in tools/libchebhip_ldj2.so spectral-petsc_amd/libchebhip.so; do
    echo "== $lib"; CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 256
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 128
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 64
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/chebmult_bench.py 64 128 256
  done
done
for lib in tools/libchebhip_ldj2.so spectral-petsc_amd/libchebhip.so; do
  echo "== $lib"; CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/stokes_bench.py
done
timeout -k 10 120 python tools/stamp_probe_multi.py 64 tools/libchebhip_diag.so
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t4.log
grep -v "^$" gpurun_out/r06_t4.log | grep -v "SQ_LDS_ADDR\|UNALIGNED\|SQ_INSTS"
