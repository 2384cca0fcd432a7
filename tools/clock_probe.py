#!/usr/bin/env python3
"""Run the matvec in a loop for a few seconds and sample rocm-smi (sclk, power) beside it.  The stream-ablated
builds of tools/v4_ablate.sh are selected with CHEBHIP_LIB_PATH (one process per build)."""
import os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as ge
sp = ge.load()
if os.environ.get("CHEBHIP_LIB_PATH"):      # diagnostic builds (tools/v4_ablate.sh): then only the first setting is meaningful
    sp.LIB_PATH = os.environ["CHEBHIP_LIB_PATH"]
P = 256
op = sp.EllipticOp((P, P, P))
U = torch.randn(op.global_size, dtype=torch.float64, device="cuda")
V = torch.empty_like(U)
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if ("sclk" in l or "Power" in l or "mclk" in l or "fclk" in l or "junction" in l.lower()) and "GPU[0]" in l]
        return " | ".join(keep)
    except Exception as e:
        return "smi failed: %r" % e
for name in (os.path.basename(os.environ.get("CHEBHIP_LIB_PATH", "shipped build")),):
    stop = False
    res = []
    def sampler():
        time.sleep(1.0)
        while not stop:
            try:
                res.append(smi())
            except Exception as e:          # a sampling failure must not end the record
                res.append("sample failed: %r" % (e,))
            time.sleep(0.7)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 5.0:
        for _ in range(200):
            op.mult(U, V)
        torch.cuda.synchronize(); n += 200
    dt = time.time() - t0
    stop = True; th.join()
    print("%-28s %.1f us/matvec" % (name, dt / n * 1e6))
    for r in res[:4]:
        print("    ", r)
