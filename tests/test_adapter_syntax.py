"""adapter/chebyshev_petsc.c through a compiler's syntax and type checking (gcc -fsyntax-only, C and C++), in both
configurations (host Vecs; -DPETSC_HAVE_HIP -DCHEBHIP_USE_DEVICE_VECS), against include/chebhip.h and against
DECLARATIONS of the PETSc entry points it calls (tests/host/petsc_decls: not PETSc, written from the manual pages).
No PETSc exists in this image, so this is as far as the binding can be taken here: it is well-formed and type-correct
against the ABI it forwards to; SURVEY 8f.2 (a live PETSc) stays open."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GCC = shutil.which("gcc")


@pytest.mark.skipif(GCC is None, reason="gcc not found")
@pytest.mark.parametrize("lang", ["c", "c++"])
@pytest.mark.parametrize("device", [False, True], ids=["host_vecs", "device_vecs"])
def test_adapter_is_well_formed(lang, device):
    cmd = [GCC, "-fsyntax-only", "-x", lang, "-Wall", "-Wextra", "-Werror=implicit-function-declaration" if lang == "c" else "-Wall",
           "-Werror=incompatible-pointer-types" if lang == "c" else "-Wall", "-Werror=int-conversion" if lang == "c" else "-Wall",
           "-I", os.path.join(ROOT, "tests", "host", "petsc_decls"), "-I", os.path.join(ROOT, "adapter"), "-I", os.path.join(ROOT, "include")]
    if device:
        cmd += ["-DPETSC_HAVE_HIP", "-DCHEBHIP_USE_DEVICE_VECS"]
    r = subprocess.run(cmd + [os.path.join(ROOT, "adapter", "chebyshev_petsc.c")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
