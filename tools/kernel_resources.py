#!/usr/bin/env python3
"""Per-kernel register / LDS / spill table of one csrc/*.hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py fused.hip [extra hipcc flags]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "spectral-petsc_amd", "csrc", sys.argv[1])
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void chebhip::", "")}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
print("%-52s %6s %6s %8s %8s %9s" % ("kernel", "VGPRs", "AGPRs", "VGPRspill", "SGPRspill", "LDS"))
for r in rows:
    print("%-52s %6s %6s %8s %8s %9s" % (r["name"][:52], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("VGPRs Spill", "?"),
                                         r.get("SGPRs Spill", "?"), r.get("LDS Size", "?")))
