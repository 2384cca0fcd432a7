#!/usr/bin/env python3
"""Marginal cost of every launch of the 128^3 power-law Stokes callbacks in the PIPELINED callback (two streams, launches back to back):
chebhip_debug_stokes_ablate leaves launches out one at a time (results are wrong, timings are not).  A profiler cannot show this: it
serialises the launches (their durations then add up to 350 us for a 276-us callback)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
sp.LIB_PATH = os.path.join(ROOT, "tools", "libchebhip_diag.so")      # `make -C spectral-petsc_amd/csrc diag`: the hook exists in diagnostic builds only
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
op = sp.StokesOp((P, P, P)); op.set_rheology(1, 1.0, 3.0, 1e-4, 1.0)
op.set_dirichlet(np.zeros(op.dirichlet_size)); op.set_force(np.zeros(op.global_size))
x = torch.randn(op.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
op.function(x, y)
L = sp.lib()
def t(fn, reps=60, warm=15):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
names = ["gather", "x/y gradient", "fused z launch", "x/y divergence", "scatter", "pressure chain"]
for fn, tag in ((lambda: op.mult(x, y), "StokesMatMult"), (lambda: op.function(x, y), "StokesFunction")):
    L.chebhip_debug_stokes_ablate(0)
    full = min(t(fn) for _ in range(3))
    print("%s: %.1f us" % (tag, full))
    for b, nm in enumerate(names):
        L.chebhip_debug_stokes_ablate(1 << b)
        v = min(t(fn) for _ in range(2))
        print("   without %-16s %.1f us  (marginal %.1f)" % (nm, v, full - v))
    for mask, nm in ((0b111111 ^ 0b000100, "ONLY the fused z launch"), (0b111111 ^ 0b001010, "ONLY the four x/y sweeps launches"), (0b111111 ^ 0b010001, "ONLY gather + scatter"),
                     (0b111111 ^ 0b100000, "ONLY the pressure chain"), (0b100000 | 0b010001, "the viscous chain without gather / scatter / pressure")):
        L.chebhip_debug_stokes_ablate(mask)
        print("   %-52s %.1f us" % (nm, min(t(fn) for _ in range(2))))
    L.chebhip_debug_stokes_ablate(0)
    if tag == "StokesMatMult":
        op.function(x, y)
