import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import __graft_entry__ as ge
sp = ge.load()
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
base = None
for it in range(6):
    op = sp.EllipticOp((130, 70, 68)); pc = sp.FdPc(op, sweeps=2)
    u = torch.rand(op.global_size, dtype=torch.float64, device="cuda") + 0.5; b = torch.randn_like(u); r = torch.empty_like(u)
    op.set_dirichlet(np.zeros(op.dirichlet_size)); op.function(u, b, r, 2.0, 2.0); op.mult(u, r); pc.update(); pc.apply(u, r)
    pc.destroy(); op.destroy()
    st = sp.StokesOp((66, 40, 36)); st.set_dirichlet(np.zeros(st.dirichlet_size)); st.set_force(np.zeros(st.global_size))
    sd = sp.StokesSaddlePc(st)
    x = torch.randn(st.global_size, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
    st.function(x, y); st.mult(x, y); sd.setup(); sd.apply(x, y)
    sd.destroy(); st.destroy()
    del u, b, r, x, y
    torch.cuda.empty_cache()
    f = free()
    if it == 1: base = f
    print(it, f >> 20, "MiB free")
assert base is not None and abs(free() - base) < (8 << 20), "device memory leak"
print("no leak")
