#!/bin/bash
# A/B builds for the prologue of cheb_sweep_vec4_kernel (V4_OVERLAP / V4_FRAG_AHEAD, sweep_vec.hip):
#   tools/libchebhip_ov0.so         the round 2-5 prologue (rotated fragment fetch, vmcnt(0) before the first tile)
#   tools/libchebhip_ldj2.so        the round 1-5 pitch of the JFAST tile image in LDS (HP + 2 doubles: 2-way bank conflicts on the operand reads)
#   tools/libchebhip_fa<N>.so       paced fetch with N pairs up front (arguments: the N to build, e.g. 2 4 8 16)
#   tools/libchebhip_diag.so        shipped kernel with in-kernel cycle stamps (-DCHEB_STAMPS)
#   tools/libchebhip_diag_ov0.so    the old prologue with stamps
#   tools/libchebhip_diag_fa<N>.so  stamps for the N given
# time them with CHEBHIP_LIB_PATH=tools/libchebhip_ov0.so tools/quick_bench.py 256 ; stamps: tools/stamp_probe3.py 256 <lib>
set -e
cd "$(dirname "$0")/../spectral-petsc_amd/csrc"
make -s
OTHERS="sweep.o sweep_xl.o fused.o fused4.o chebhip.o stokes.o krylov.o diffmat.o precond.o saddle.o dist.o comm.o slabx.o options.o"
F="-O3 -fPIC -std=c++17 --offload-arch=gfx950"
build() { # name flags...
  local name=$1; shift
  /opt/rocm/bin/hipcc $F "$@" -c sweep_vec.hip -o /tmp/sweep_vec_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libchebhip_$name.so /tmp/sweep_vec_$name.o $OTHERS -ldl
}
build ov0 -DV4_OVERLAP=0 &
build ldj2 -DV_LDJ_PAD=2 &
build diag -DCHEB_STAMPS &
build diag_ov0 -DCHEB_STAMPS -DV4_OVERLAP=0 &
wait
for n in "$@"; do
  build fa$n -DV4_FRAG_AHEAD=$n &
  build diag_fa$n -DCHEB_STAMPS -DV4_FRAG_AHEAD=$n &
  wait
done
