#!/usr/bin/env python3
"""Diagnostic only: run the Poisson matvec with tools/libchebhip_diag.so (sweep_vec.hip built with
-DCHEB_STAMPS) and print where a wave's cycles go per sub-tile: pre-chain, chain, post-chain, barrier."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, "tools", "libchebhip_diag.so")
L = sp.lib()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
for ab in (0, 3):
    L.chebhip_debug_ablate(ab)
    L.chebhip_debug_stamp_buffer(C.c_void_p(buf.data_ptr()))
    for _ in range(50):
        op.mult(U, V)
    torch.cuda.synchronize()
    raw = buf.cpu().numpy().reshape(256, 8, 8)
    b = raw[:, :, :4].astype(float)   # last launch (direction 2) of the last matvec
    nsub = float(os.environ.get("NSUB", "16"))
    print("ablate=%d  cycles per sub-tile per wave (mean over waves; waves 0-3 / 4-7):" % ab)
    for k, name in enumerate(("pre-chain", "chain", "post-chain", "barrier")):
        div = nsub if k < 3 else nsub / 2
        print("   %-10s  all %8.0f   A %8.0f   B %8.0f" % (name, b[:, :, k].mean() / div, b[:, :4, k].mean() / div, b[:, 4:, k].mean() / div))
    tot = b.sum(axis=2).mean() / nsub
    print("   total per sub-tile %8.0f cycles (s_memtime ticks)" % tot)
    print("   prologue %8.0f  loop %8.0f  (mean per wave);  kernel span first-begin..last-end %8.0f;  begin spread %8.0f  end spread %8.0f" % (
        raw[:, :, 4].mean(), raw[:, :, 5].mean(), float(raw[:, :, 7].max() - raw[:, :, 6].min()), float(raw[:, :, 6].max() - raw[:, :, 6].min()), float(raw[:, :, 7].max() - raw[:, :, 7].min())))
    print("   loop cycles by block: min %8.0f max %8.0f" % (raw[:, :, 5].min(), raw[:, :, 5].max()))
L.chebhip_debug_stamp_buffer(None)
