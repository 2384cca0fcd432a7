"""cheb_apply against the reference's transform recipe run on the vendor FFT (tests/fft_recipe.py): the HIP path is a dense
product, the reference an FFT-based one -- this is the comparison of the two algorithms on the device itself, on N(0,1)
inputs at sizes up to BASELINE config 3, in every direction.  Tolerance 1e-11 normwise (observed 1e-15 .. 1e-13; the FFT
route loses a few digits to the division by sin(theta) near the end points)."""
import numpy as np
import pytest
import torch

import __graft_entry__ as ge
from fft_recipe import cheb_fft

pytestmark = pytest.mark.gpu
sp = ge.load()


@pytest.mark.parametrize("shape", [(64, 64, 64), (128, 128, 128), (33, 20, 17), (200, 130), (7, 256, 12), (256, 256, 256), (100,)],
                         ids=lambda s: "x".join(map(str, s)))
def test_cheb_apply_equals_the_fft_recipe(shape):
    torch.manual_seed(20240229 + len(shape))
    x = torch.randn(shape, dtype=torch.float64, device="cuda")
    y = torch.empty_like(x)
    for tr in range(len(shape)):
        if shape[tr] < 3:
            continue
        plan = sp.ChebPlan(shape, tr)
        plan.mult(x.reshape(-1), y.reshape(-1))
        ref = cheb_fft(x, tr)
        torch.cuda.synchronize()
        err = float((ref - y).norm() / ref.norm())
        plan.destroy()
        assert err < 1e-11, (shape, tr, err)
