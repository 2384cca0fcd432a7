#!/bin/bash
# Where does cheb_sweep_xl_kernel spend its time?  Builds variants of sweep_xl.hip IN THE GPU BOX'S COPY of the tree (run it
# through gpurun: the snapshot is thrown away afterwards) and times ChebMult on 1024 x 8192 (dim 0) and 8192 x 1024 (dim 1):
#   base        the kernel as it is
#   nofrag      the matrix stream switched off (the first fragments are reused): image side + MFMA only  -- results are wrong
#   noimg       the image loads switched off (LDS filled with constants): matrix stream + MFMA only      -- results are wrong
#   core        both off: MFMA chains + LDS operand reads + result stores                                   -- results are wrong
#   mfma        ... and the LDS operand reads replaced by register values: the MFMA chains alone            -- results are wrong
# Measured (profiles/r03_xl_ablate.txt, 1024 x 8192): base 174 us (49 TF), nofrag 153, noimg 163, core 140 (61 TF), mfma 141-145:
# the chains run at 0.77 of peak launch included, the LDS operand reads are free, the two streams cost 8 % and 14 %.
# usage: tools/xl_ablate.sh   (prints one line per variant)
set -e
cd "$(dirname "$0")/.."
SRC=spectral-petsc_amd/csrc/sweep_xl.hip
cp $SRC /tmp/sweep_xl.orig
run() {
  make -C spectral-petsc_amd/csrc -s -j8 >/dev/null 2>&1
  python3 - "$1" <<'PY'
import sys, os, torch
sys.path.insert(0, os.getcwd())
import __graft_entry__ as ge
sp = ge.load()
out = []
for shape, tr in (((1024, 8192), 0), ((8192, 1024), 1), ((384, 384, 384), 1)):
    x = torch.randn(shape, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    plan = sp.ChebPlan(shape, tr)
    for _ in range(5): plan.mult(x, y)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): plan.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    out.append("%s tr=%d %.1f us (%.1f TF)" % ("x".join(map(str, shape)), tr, us, shape[tr] * x.numel() / us / 1e6))
    plan.destroy()
print("%-8s %s" % (sys.argv[1], "   ".join(out)), flush=True)
PY
}
run base
sed -i 's|if (u == 0 \|\| live\[u\]) { ae\[u\]\[s\] = fE\[u\]\[(long)nx \* 64\]; ao\[u\]\[s\] = fO\[u\]\[(long)nx \* 64\]; }|asm volatile("" :: "s"(nx));|' $SRC; run nofrag; cp /tmp/sweep_xl.orig $SRC
sed -i 's|va\[it\] = ok ? p.in0\[a + (u32)j \* inner\] : 0.0;|va[it] = ok ? 1.0 : 0.0;|; s|vb\[it\] = (ok \&\& 2 \* j != nn) ? p.in0\[a + (u32)(nn - j) \* inner\] : 0.0;|vb[it] = ok ? 0.5 : 0.0;|' $SRC; run noimg; cp /tmp/sweep_xl.orig $SRC
# both streams off: MFMA + LDS operand reads + result stores only
sed -i 's|if (u == 0 \|\| live\[u\]) { ae\[u\]\[s\] = fE\[u\]\[(long)nx \* 64\]; ao\[u\]\[s\] = fO\[u\]\[(long)nx \* 64\]; }|asm volatile("" :: "s"(nx));|; s|va\[it\] = ok ? p.in0\[a + (u32)j \* inner\] : 0.0;|va[it] = ok ? 1.0 : 0.0;|; s|vb\[it\] = (ok \&\& 2 \* j != nn) ? p.in0\[a + (u32)(nn - j) \* inner\] : 0.0;|vb[it] = ok ? 0.5 : 0.0;|' $SRC; run core; cp /tmp/sweep_xl.orig $SRC
# ... and without the LDS operand reads as well (constants): the MFMA chains alone
sed -i 's|if (u == 0 \|\| live\[u\]) { ae\[u\]\[s\] = fE\[u\]\[(long)nx \* 64\]; ao\[u\]\[s\] = fO\[u\]\[(long)nx \* 64\]; }|asm volatile("" :: "s"(nx));|; s|va\[it\] = ok ? p.in0\[a + (u32)j \* inner\] : 0.0;|va[it] = ok ? 1.0 : 0.0;|; s|vb\[it\] = (ok \&\& 2 \* j != nn) ? p.in0\[a + (u32)(nn - j) \* inner\] : 0.0;|vb[it] = ok ? 0.5 : 0.0;|; s|const double be = imgE\[bi\], bo = imgO\[bi\];|double be = (double)bi, bo = 1.0 - be; asm volatile("" : "+v"(be), "+v"(bo));|' $SRC; run mfma; cp /tmp/sweep_xl.orig $SRC
make -C spectral-petsc_amd/csrc -s -j8 >/dev/null 2>&1
