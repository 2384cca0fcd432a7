#!/usr/bin/env python3
"""Why does bench.py read 292-295 / 320-323 us for the 128^3 power-law Stokes callbacks when tools/stokes_ab.py reads 280 / 296 on the
same sources?  One process: handles made in a fresh allocator state, the same handles re-timed after
gigabytes have been allocated and freed (torch tensors, Krylov bases, a whole config-5 solve as bench.py's solves() runs it), and
handles made afterwards.  If an OLD handle slows down it is the process state (clocks, TLB, fragmentation of the page tables);
if only NEW handles are slow it is where their arrays land."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
POWER = (1, 1.0, 3.0, 1e-4, 1.0)


def make():
    op = sp.StokesOp((P, P, P)); op.set_rheology(*POWER)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
    return op


def t(fn, reps=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


x = None
shown = set()
def timed(tag, op):
    global x
    if x is None:
        x = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    f = [t(lambda: op.function(x, y)) for _ in range(3)]
    m = [t(lambda: op.mult(x, y)) for _ in range(3)]
    free, tot = torch.cuda.mem_get_info()
    print("%-34s Function %s  MatMult %s   (free %.1f GB)" % (tag, " ".join("%.1f" % v for v in f), " ".join("%.1f" % v for v in m), free / 2**30), flush=True)
    if id(op) not in shown:
        shown.add(id(op))
        import ctypes as C
        a = (C.c_ulonglong * 32)()
        L = sp.lib(); L.chebhip_debug_stokes_arrays.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        n = L.chebhip_debug_stokes_arrays(op._h, 32, a)
        base = min(v for v in a[:n] if v)
        names = "xL yL xF yLx1 yLx2 V0 S0 gp0 V1 S1 gp1 V2 S2 gp2 eta deta T pL p2".split()
        print("   " + " ".join("%s+%.0f" % (names[i], (a[i] - base) / 2**20) for i in range(n) if a[i]) + " MiB (base %#x)" % base, flush=True)


if len(sys.argv) > 2:
    for kv in sys.argv[2:]:
        k, v = kv.split("="); sp.set_option(k, int(v))
    print("options:", sys.argv[2:])
A = make(); timed("A fresh", A)
B = make(); timed("B fresh (second handle)", B); timed("A again", A)
big = [torch.empty(2 * 2**30 // 8, dtype=torch.float64, device="cuda").normal_() for _ in range(6)]
del big; torch.cuda.empty_cache()
timed("A after 12 GB torch churn", A)
C = make(); timed("C made after torch churn", C)
# what bench.py's solves() leaves behind: an operator of 256^3, a preconditioner, Krylov bases of gigabytes -- made and destroyed
op = sp.EllipticOp((256, 256, 256)); op.set_dirichlet(np.zeros(op.dirichlet_size))
us = torch.rand(op.global_size, dtype=torch.float64, device="cuda"); b = torch.empty_like(us)
op.function(us, None, b, 4.0, 2.0)
pc = sp.FdPc(op, sweeps=0); pc.update()
ks = sp.Fgmres(op.global_size, restart=30); ks2 = sp.Fgmres(8001504, restart=60)
torch.cuda.synchronize(); ks.destroy(); ks2.destroy(); pc.destroy(); op.destroy(); del us, b
timed("A after the 256^3 elliptic solve", A)
D = make(); timed("D made after the elliptic solve", D)
st = make(); pcs = sp.FdPc(st, sweeps=0); M = sp.StokesSaddlePc(st, 0); M.setup()
ws = torch.randn(st.global_size, dtype=torch.float64, device="cuda"); zs = torch.empty_like(ws)
M.apply(ws, zs); torch.cuda.synchronize(); M.destroy(); pcs.destroy(); st.destroy(); del ws, zs
timed("A after a Stokes solve", A)
E = make(); timed("E made after the Stokes solve", E)
timed("B again", B); timed("C again", C)
