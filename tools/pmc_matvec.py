#!/usr/bin/env python3
"""A short loop of Poisson matvecs for rocprofv3 counter passes: usage pmc_matvec.py [P] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
for i in range(n):
    op.mult(U, V)
torch.cuda.synchronize()
print("done", P, n)
