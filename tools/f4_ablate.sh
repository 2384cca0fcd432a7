#!/bin/bash
# Diagnostic builds of libchebhip.so with one stream of cheb_fused4_kernel removed (F4_ABLATE bits, see fused4.hip):
# tools/f4_ablate.sh 1 2 4 8   ->  tools/libchebhip_f4a<bits>.so ; time them with CHEBHIP_LIB_PATH=... tools/elliptic_bench.py 256
set -e
cd "$(dirname "$0")/../spectral-petsc_amd/csrc"
make -s
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DF4_ABLATE=$b -c fused4.hip -o /tmp/fused4_a$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libchebhip_f4a$b.so sweep.o sweep_xl.o sweep_vec.o fused.o /tmp/fused4_a$b.o chebhip.o stokes.o krylov.o diffmat.o precond.o saddle.o dist.o comm.o slabx.o options.o -ldl
done
