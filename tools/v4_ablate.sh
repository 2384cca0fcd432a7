#!/bin/bash
# Diagnostic builds of libchebhip.so with one stream of cheb_sweep_vec4_kernel removed (V4_ABLATE bits, see sweep_vec.hip):
# tools/v4_ablate.sh 1 2 8 11 16  ->  tools/libchebhip_v4a<bits>.so ; time them with CHEBHIP_LIB_PATH=... tools/quick_bench.py 256
set -e
cd "$(dirname "$0")/../spectral-petsc_amd/csrc"
make -s
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DV4_ABLATE=$b -c sweep_vec.hip -o /tmp/sweep_vec_a$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libchebhip_v4a$b.so sweep.o sweep_xl.o /tmp/sweep_vec_a$b.o fused.o fused4.o chebhip.o stokes.o krylov.o diffmat.o precond.o saddle.o dist.o comm.o slabx.o options.o -ldl
done
