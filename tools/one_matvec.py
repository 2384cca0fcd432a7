import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
for i in range(4):
    torch.cuda.synchronize(); t = time.time()
    op.mult(U, V)
    torch.cuda.synchronize()
    print("P=%d matvec %d: %.3f ms" % (P, i, (time.time() - t) * 1e3), flush=True)
