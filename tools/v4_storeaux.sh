#!/bin/bash
# Diagnostic builds of libchebhip.so with a cache policy on the result stores of cheb_sweep_vec4_kernel (V4_STORE_AUX: 1 = sc0,
# 2 = nt, 16 = sc1, 17 = sc0 sc1, 18 = sc1 nt): tools/v4_storeaux.sh 16 17 2 -> tools/libchebhip_v4s<aux>.so
set -e
cd "$(dirname "$0")/../spectral-petsc_amd/csrc"
make -s
for b in "$@"; do
  /opt/rocm/bin/hipcc -O3 -fPIC -std=c++17 --offload-arch=gfx950 -DV4_STORE_AUX=$b -c sweep_vec.hip -o /tmp/sweep_vec_s$b.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libchebhip_v4s$b.so sweep.o sweep_xl.o /tmp/sweep_vec_s$b.o fused.o fused4.o chebhip.o stokes.o krylov.o diffmat.o precond.o saddle.o dist.o comm.o slabx.o options.o -ldl
done
