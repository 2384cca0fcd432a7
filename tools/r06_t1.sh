set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r06_t1_parity.log 2>&1 || { tail -30 gpurun_out/r06_t1_parity.log; exit 1; }
tail -3 gpurun_out/r06_t1_parity.log
{
for rep in 1 2; do
  for lib in tools/libchebhip_ov0.so spectral-petsc_amd/libchebhip.so; do
    echo "== $lib"; CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 256
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 120 python tools/quick_bench.py 128
  done
done
for lib in tools/libchebhip_ov0.so spectral-petsc_amd/libchebhip.so; do
  echo "== $lib"
  for G in 8 4; do CHEBHIP_LIB_PATH=$lib timeout -k 10 120 python tools/dist_rank_trace.py $G 200 dist_single_stream=1; CHEBHIP_LIB_PATH=$lib timeout -k 10 120 python tools/dist_rank_trace.py $G 200; done
done
timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag_ov0.so
timeout -k 10 120 python tools/stamp_probe3.py 256 tools/libchebhip_diag.so
timeout -k 10 120 python tools/stamp_probe3.py 128 tools/libchebhip_diag_ov0.so
timeout -k 10 120 python tools/stamp_probe3.py 128 tools/libchebhip_diag.so
} > gpurun_out/r06_t1_ab.log 2>&1
cat gpurun_out/r06_t1_ab.log
