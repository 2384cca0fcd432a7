#!/bin/bash
# The record behind "the headline kernel runs at the package's power cap" (DESIGN 4.2b), taken in one go on the GPU box:
#   1. in-kernel clock of cheb_sweep_vec4_kernel (diagnostic build with stamps: d s_memtime / d s_memrealtime x 100 MHz,
#      after >= 2 s of back-to-back launches on random data) -- the primary evidence (MI355X_MICROARCH.md, DVFS note 6);
#   2. rocm-smi package power beside 5-s loops of the shipped build, of the build without global memory traffic
#      (V4_ABLATE=11) and of the build without MFMA chains (V4_ABLATE=16) -- one process each, stderr kept.
# usage (on the GPU box): tools/power_clock_record.sh > gpurun_out/r03_power_clock.txt 2>&1
set -e
cd "$(dirname "$0")/.."
make -s -C spectral-petsc_amd/csrc diag
tools/v4_ablate.sh 11 16
echo "== in-kernel clock (tools/stamp_probe3.py, diagnostic build)"
python3 tools/stamp_probe3.py 256 2>&1 | grep -v amdgpu.ids
echo "== shipped build"
python3 tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
for b in 11 16; do
  echo "== V4_ABLATE=$b ($([ $b = 11 ] && echo 'no global memory traffic' || echo 'no MFMA chains'))"
  CHEBHIP_LIB_PATH=tools/libchebhip_v4a$b.so python3 tools/clock_probe.py 2>&1 | grep -v amdgpu.ids
done
