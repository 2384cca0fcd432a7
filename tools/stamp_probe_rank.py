#!/usr/bin/env python3
"""Diagnostic only: in-kernel stamps of the one launch that runs the three directions of a slab rank's matvec (cheb_sweep_multi_gather_kernel:
jobs 0, 1 = the local directions, job 2 = the pencil direction over the ranks' slabs; NULL transport, stamped build).
usage: stamp_probe_rank.py [G] [lib.so]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sp = ge.load(); dsp = ge.load_dist()
sp.LIB_PATH = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "libchebhip_diag.so")
L = sp.lib()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
comm = dsp.Comm(sp, null=(G, 0))
D = dsp.DistPoissonC((256, 256, 256), sp, comm=comm)
U = torch.randn(D.local_size, dtype=torch.float64, device="cuda"); V = torch.empty_like(U)
for _ in range(300):
    D.mult(U, V)
torch.cuda.synchronize()
AREA0, AREA = 3 * 256 * 8 * 16, 512 * 8 * 16
buf = torch.zeros(AREA0 + 18 * AREA, dtype=torch.int64, device="cuda")
L.chebhip_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(6):
    D.mult(U, V)
torch.cuda.synchronize()
L.chebhip_debug_stamp_buffer(None)
raw = buf.cpu().numpy()[AREA0:].reshape(18, 512, 8, 16).astype(float)
names = ("chain0", "epi0", "chain1", "epi1", "barrier")
print("library %s, one rank of %d at 256^3: the launch of the three directions (slots 0 / 9: direction 1, 1 / 10: direction 2, 2 / 11: the pencil direction)" % (os.path.basename(sp.LIB_PATH), G))
for j in range(18):
    r = raw[j]
    live = r[:, :, 6] > 0
    if not live.any():
        continue
    nwg = int(live[:, 0].sum())
    cyc, ticks = r[:, :, 6][live], r[:, :, 7][live]
    ghz = np.median(cyc / np.maximum(ticks, 1.0)) * 0.1
    print("  slot %2d: %3d workgroups; prologue %6.0f  first tile done %6.0f  whole kernel %7.0f cycles (max %7.0f) = %.1f us at %.2f GHz" % (
        j, nwg, r[:, :, 5][live].mean(), r[:, :, 10][live].mean(), cyc.mean(), cyc.max(), cyc.mean() / ghz / 1e3, ghz))
    print("           per workgroup, sums over its tiles: " + "  ".join("%s %6.0f / %6.0f" % (nm, r[:, :4, k][live[:, :4]].mean(), r[:, 4:, k][live[:, 4:]].mean()) for k, nm in enumerate(names)) + "   (waves 0-3 / 4-7)")
