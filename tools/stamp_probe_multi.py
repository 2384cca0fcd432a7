#!/usr/bin/env python3
"""Diagnostic only: where the waves of the multi-job sweep launches of a 64^3 Stokes callback spend their cycles (stamped build:
tools/v4_overlap_ab.sh / `make diag`; cheb_sweep_multi_kernel -> vec1_body stamps).  Per job of the LAST multi-job launch issued
(a StokesMatMult runs two: nine jobs, then the three grad div v jobs -- the probe stamps them one at a time by call order):
fragments landed, loop entered, per-tile pre / chain / post / barrier, whole kernel, tiles walked.
usage: stamp_probe_multi.py [P] [lib.so]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "libchebhip_diag.so")
L = sp.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
op = sp.StokesOp((P, P, P))
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
for _ in range(200):
    op.mult(x, y)
torch.cuda.synchronize()
AREA0, AREA = 3 * 256 * 8 * 16, 512 * 8 * 16
buf = torch.zeros(AREA0 + 18 * AREA, dtype=torch.int64, device="cuda")
L.chebhip_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(5):
    op.mult(x, y)
torch.cuda.synchronize()
L.chebhip_debug_stamp_buffer(None)
raw = buf.cpu().numpy()[AREA0:].reshape(18, 512, 8, 16).astype(float)
print("library %s, Stokes %d^3 StokesMatMult: the stamps are those of the LAST multi-job launch that used each job slot" % (os.path.basename(sp.LIB_PATH), P))
print("(slots 0-8: the nine-job launch -- 0-2 second-derivative sweeps of 3 fields each, 3-5 trace, 6-8 pressure gradient; 9-11: the grad div v launch)")
for j in range(18):
    r = raw[j]
    live = r[:, :, 11] > 0
    if not live.any():
        continue
    nt = np.maximum(r[:, :, 9][live], 1.0)
    ghz = np.median(r[:, :, 11][live] / np.maximum(r[:, :, 10][live], 1.0)) * 0.1
    print("  slot %d: %3d workgroups, tiles per workgroup %.2f (max %d); first tiles and fragments landed %6.0f  loop entered %6.0f  whole kernel %6.0f cycles (max %6.0f) = %.1f us at %.2f GHz" % (
        j, live[:, 0].sum(), nt.mean(), nt.max(), r[:, :, 8][live].mean(), (r[:, :, 8] + r[:, :, 4])[live].mean(), r[:, :, 11][live].mean(), r[:, :, 11][live].max(),
        r[:, :, 11][live].mean() / ghz / 1e3, ghz))
    print("          per tile: pre %6.0f  chain %6.0f  post %6.0f  barrier %6.0f" % tuple((r[:, :, k][live] / nt).mean() for k in range(4)))
