#!/usr/bin/env python3
"""Diagnostic only: the Poisson matvec with a stamped build of the library (tools/v4_overlap_ab.sh or
`make -C spectral-petsc_amd/csrc diag`: sweep_vec.hip with -DCHEB_STAMPS) -- where a wave of cheb_sweep_vec4_kernel
spends its cycles, per launch (direction) and wave group: chain 0, epilogue 0, chain 1, epilogue 1, barrier; the
prologue split (fragments requested / landed, first lines parked, loop entered, first tile done); the spread of
the waves' begin and end stamps over the launch (dispatch skew and tail).
usage: stamp_probe3.py [P] [lib.so] [slab planes (dimension 0 of a Px(P)x(P) slab; default P)]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, sys.argv[2]) if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "libchebhip_diag.so")
L = sp.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
W = 16
buf = torch.zeros(3 * 256 * 8 * W, dtype=torch.int64, device="cuda")
names = ("chain0", "epi0", "chain1", "epi1/top", "barrier")
import time
t0 = time.time()
while time.time() - t0 < 2.5:                      # >= 2 s of back-to-back launches on random data before the stamped ones
    for _ in range(200):
        op.mult(U, V)
    torch.cuda.synchronize()
L.chebhip_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
for _ in range(9):
    op.mult(U, V)
torch.cuda.synchronize()
L.chebhip_debug_stamp_buffer(None)
raw = buf.cpu().numpy().reshape(3, 256, 8, W).astype(float)
print("library %s, P = %d" % (os.path.basename(sp.LIB_PATH), P))
for k in range(3):
    r = raw[k]
    live = r[:, :, 6] > 0
    nt = max(1.0, np.ceil((P - 2) ** 2 / 32.0 / 256.0)) if P > 64 else 1.0
    cyc, ticks = r[:, :, 6][live], r[:, :, 7][live]
    ghz = np.median(cyc / np.maximum(ticks, 1.0)) * 0.1
    print("  launch %d: prologue %7.0f  loop %7.0f  whole kernel %7.0f shader cycles in %6.1f us of s_memrealtime -> in-kernel clock %.3f GHz (median over waves)" % (
        k, r[:, :, 5][live].mean(), r[:, :, :5].sum(axis=2)[live].mean(), cyc.mean(), ticks.mean() / 100.0, ghz))
    print("     prologue split: fragment requests issued / set landed %6.0f   first lines parked %6.0f   loop entered %6.0f   first tile done %6.0f" % (
        r[:, :, 8][live].mean(), r[:, :, 9][live].mean(), r[:, :, 5][live].mean(), r[:, :, 10][live].mean()))
    # s_memtime counts per XCD (the XCDs' counters are not aligned with each other): spans are taken per XCD = BID % 8
    spans, bsp, esp = [], [], []
    for x in range(8):
        m = live[x::8]
        if m.any():
            b, e = r[x::8, :, 11][m], r[x::8, :, 12][m]
            spans.append(e.max() - b.min()); bsp.append(b.max() - b.min()); esp.append(e.max() - e.min())
    print("     per XCD: waves begin over %6.0f cycles, end over %6.0f; first begin -> last end %7.0f cycles (= %.1f us at the in-kernel clock), means over the XCDs" % (
        np.mean(bsp), np.mean(esp), np.mean(spans), np.mean(spans) / ghz / 1e3))
    for s_, nm in enumerate(names):
        print("     %-9s per tile:  waves 0-3 %7.0f   waves 4-7 %7.0f" % (nm, r[:, :4, s_].mean() / nt, r[:, 4:, s_].mean() / nt))
