#!/usr/bin/env python3
"""Diagnostic only: the Poisson matvec with tools/libchebhip_diag.so (`make -C spectral-petsc_amd/csrc diag`:
sweep_vec.hip with -DCHEB_STAMPS) -- where a wave of cheb_sweep_vec3_kernel spends its cycles, per launch
(direction) and wave group: chain 0, epilogue 0, chain 1, epilogue 1, barrier; prologue; spans.
usage: stamp_probe3.py [P]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, "tools", "libchebhip_diag.so")
L = sp.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
variants = [0]
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
buf = torch.zeros(3 * 256 * 8 * 8, dtype=torch.int64, device="cuda")
names = ("chain0", "epi0", "chain1", "epi1/top", "barrier")
for var in variants:
    import time
    t0 = time.time()
    while time.time() - t0 < 2.5:                      # >= 2 s of back-to-back launches on random data before the stamped ones
        for _ in range(200):
            op.mult(U, V)
        torch.cuda.synchronize()
    L.chebhip_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(10):
        op.mult(U, V)
    torch.cuda.synchronize()
    L.chebhip_debug_stamp_buffer(None)
    raw = buf.cpu().numpy().reshape(3, 256, 8, 8).astype(float)
    print("variant %d" % var)
    for k in range(3):
        r = raw[k]
        ntile = 8.0
        cyc, ticks = r[:, :, 6], r[:, :, 7]
        ghz = np.median(cyc / np.maximum(ticks, 1.0)) * 0.1
        print("  launch %d: prologue %7.0f  loop %7.0f  whole kernel %7.0f shader cycles in %6.1f us of s_memrealtime -> in-kernel clock %.3f GHz (median over waves)" % (
            k, r[:, :, 5].mean(), r[:, :, :5].sum(axis=2).mean(), cyc.mean(), ticks.mean() / 100.0, ghz))
        for s_, nm in enumerate(names):
            print("     %-9s per tile:  waves 0-3 %7.0f   waves 4-7 %7.0f" % (nm, r[:, :4, s_].mean() / ntile, r[:, 4:, s_].mean() / ntile))
