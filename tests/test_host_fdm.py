"""Host logic of the finite-difference preconditioner: the fast-diagonalisation data of csrc/diffmat.cpp.
The end modes of a Gauss-Lobatto line come in even / odd pairs that are degenerate to far below rounding; the parity
classes are diagonalised separately and laid out by parity (the layout the raw modes of the sweep kernels rely on).
Compiles a small harness with hipcc (host code only); no GPU."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not found")
def test_fdm_line_decomposition(tmp_path):
    csrc = os.path.join(ROOT, "spectral-petsc_amd", "csrc")
    objs = []
    for src, extra in ((os.path.join(csrc, "diffmat.cpp"), ["-x", "hip"]), (os.path.join(csrc, "options.cpp"), ["-x", "hip"]),
                       (os.path.join(ROOT, "tests", "host", "fdm_check.cpp"), [])):
        o = str(tmp_path / (os.path.basename(src) + ".o"))
        subprocess.run([HIPCC, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", csrc] + extra + ["-c", src, "-o", o], check=True, timeout=600)
        objs.append(o)
    exe = str(tmp_path / "fdm_check")
    subprocess.run([HIPCC] + objs + ["-o", exe], check=True, timeout=600)
    sizes = [3, 4, 5, 8, 9, 33, 64, 66, 130, 255, 256, 258]
    out = subprocess.run([exe] + [str(p) for p in sizes], check=True, capture_output=True, text=True, timeout=600).stdout.split("\n")
    rows = [l.split() for l in out if l.strip()]
    assert [int(r[0]) for r in rows] == sizes
    for r in rows:
        assert r[1] != "failed", r
        e_inv, e_eig, e_par, ok = float(r[1]), float(r[2]), float(r[3]), int(r[4])
        assert e_inv < 1e-16 and e_eig < 1e-16 and e_par < 1e-16 and ok == 1, r
