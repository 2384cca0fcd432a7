set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stokes.py tests/test_gpu_precond.py tests/test_gpu_identities.py -x -q -m gpu > gpurun_out/r06_t10_tests.log 2>&1 || { tail -40 gpurun_out/r06_t10_tests.log; exit 1; }
tail -3 gpurun_out/r06_t10_tests.log
{
for rep in 1 2; do
  for lib in tools/libchebhip_pad2.so spectral-petsc_amd/libchebhip.so; do
    echo "== $lib"
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/stokes_bench.py
    CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/elliptic_bench.py 256
  done
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t10.log
cat gpurun_out/r06_t10.log
