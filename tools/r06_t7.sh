set -e
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
{
for P in 192 200 208 224 240 256; do
  timeout -k 10 120 python tools/quick_bench.py $P poisson_launches=2
  timeout -k 10 120 python tools/quick_bench.py $P poisson_launches=3
done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_t7.log
cat gpurun_out/r06_t7.log
