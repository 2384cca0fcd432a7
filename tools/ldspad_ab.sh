#!/bin/bash
# A/B build for the LDS row pitch of the contiguous-line tilings outside sweep_vec.hip (round 6: odd pitches): tools/libchebhip_pad2.so =
# k_st_zfused16, k_fdm_zsolve16 and cheb_fused4_kernel with the pitch = 2 mod 32 of rounds 2-5 (ZF_PAD / FZ_PAD / F4_LDJ_PAD = 2).
# time with CHEBHIP_LIB_PATH=tools/libchebhip_pad2.so tools/stokes_bench.py | tools/elliptic_bench.py 256 | tools/solve_ab.py
set -e
cd "$(dirname "$0")/../spectral-petsc_amd/csrc"
make -s
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950"
/opt/rocm/bin/hipcc $F -DZF_PAD=2 -c stokes.hip -o /tmp/stokes_pad2.o &
/opt/rocm/bin/hipcc $F -DFZ_PAD=2 -c precond.hip -o /tmp/precond_pad2.o &
/opt/rocm/bin/hipcc $F -DF4_LDJ_PAD=2 -c fused4.hip -o /tmp/fused4_pad2.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libchebhip_pad2.so sweep.o sweep_xl.o sweep_vec.o fused.o /tmp/fused4_pad2.o chebhip.o /tmp/stokes_pad2.o krylov.o diffmat.o /tmp/precond_pad2.o saddle.o dist.o comm.o slabx.o options.o -ldl
