set -e
export PYTHONUNBUFFERED=1
for rep in 1 2 3; do
  for lib in tools/libchebhip_un2.so spectral-petsc_amd/libchebhip.so; do
    echo "== $lib"; CHEBHIP_LIB_PATH=$PWD/$lib timeout -k 10 200 python tools/stokes_bench.py 2>&1 | grep "64^3 linear"
  done
done
