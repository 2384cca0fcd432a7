"""The reference's ChebMult recipe -- DCT-I -> times k -> DST-I -> scale, chebyshev.c:157-193 -- evaluated with the vendor FFT
(torch.fft = rocFFT on the real even / odd extensions of length 2(P-1)) and elementwise torch ops.  A third restatement of
the same arithmetic, independent both of the HIP kernels (dense parity-split products) and of the CPU oracle (its own
transforms): tests/test_gpu_fft_route.py checks cheb_apply against it, tools/fft_route_bench.py times it."""
import math

import torch


def cheb_fft(x, dim):
    P = x.shape[dim]; n = P - 1
    shape = [1] * x.dim(); shape[dim] = -1
    k = torch.arange(0, n + 1, dtype=torch.float64, device=x.device).view(shape)
    xe = torch.cat([x, x.flip(dim).narrow(dim, 1, n - 1)], dim)                 # even extension, length 2n
    Y = torch.fft.rfft(xe, dim=dim).real                                        # REDFT00: Y_0..Y_n            (:157)
    W = Y * k                                                                   # k * Y_k                      (:171)
    Wi = W.narrow(dim, 1, n - 1)
    sgn = torch.where(torch.arange(0, n + 1, device=x.device) % 2 == 0, 1.0, -1.0).to(torch.float64).view(shape)
    y0 = (W * k).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * n * Y.narrow(dim, n, 1)                       # (:172,176)
    yn = ((W * k) * (-sgn)).narrow(dim, 1, n - 1).sum(dim, keepdim=True) / n + 0.5 * (-1.0) ** (n + 1) * n * Y.narrow(dim, n, 1)   # (:173,177)
    z = torch.zeros_like(x.narrow(dim, 0, 1))
    oe = torch.cat([z, Wi, z, -Wi.flip(dim)], dim)                              # odd extension, length 2n
    Z = -torch.fft.rfft(oe, dim=dim).imag.narrow(dim, 1, n - 1)                 # RODFT00: 2 sum W_k sin(pi j k / n)  (:181)
    j = torch.arange(1, n, dtype=torch.float64, device=x.device).view(shape)
    yi = Z / (2.0 * n * torch.sin(math.pi * j / n))                             # (:190)
    return torch.cat([y0, yi, yn], dim)
