"""Tucker-factorised spectral weights (TFNO): host side of dlwp_cmode_product.

Reference call site: neuralop.models.TFNO(..., rank=rank) at src/dlwpbench/models/fno/fno.py:136-146
(factorization="Tucker"; tltorch/tensorly are third-party and absent: semantics restated from their published
algorithm, SURVEY.md App. A-1, parity unpinned).  W[i,o,x,y] = sum core[a,b,c,d] U_i[i,a] U_o[o,b] U_x[x,c] U_y[y,d],
all complex.  The dense weight is rebuilt once per optimizer step and written into the flat parameter buffer in
the mode-major layout the spectral kernels read; its gradient flows back to the factors through the same
kernels (dlwp_cmode_product_bwd).
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import lib as L


def tucker_rank(shape, rank):
    """tensorly.tucker_tensor.validate_tucker_rank for a float `rank` (fraction of the dense parameter count):
    solve  sum_i s_i^2 c + prod(s) c^n = rank * prod(s)  for c, then r_i = max(round(c s_i), 1)."""
    if isinstance(rank, (list, tuple)):
        return [int(r) for r in rank]
    if isinstance(rank, int) and not isinstance(rank, bool):
        return [min(int(rank), s) for s in shape]
    n = len(shape)
    target = float(np.prod(shape)) * float(rank)
    sq = float(sum(s * s for s in shape))
    full = float(np.prod(shape))
    f = lambda c: target - sq * c - full * c ** n
    lo, hi = 0.0, max(float(rank), 1.0)
    for _ in range(200):                      # bisection (brentq in tensorly)
        mid = 0.5 * (lo + hi)
        if f(mid) > 0:
            lo = mid
        else:
            hi = mid
    c = 0.5 * (lo + hi)
    return [max(int(round(s * c)), 1) for s in shape]


class _ModeProduct(torch.autograd.Function):
    """out = x x_axis U   (x complex stored as [..., 2]; U [N, R, 2])"""

    @staticmethod
    def forward(ctx, x, U, axis):
        shape = list(x.shape[:-1])
        O = int(np.prod(shape[:axis])) if axis > 0 else 1
        R = shape[axis]
        I = int(np.prod(shape[axis + 1:])) if axis + 1 < len(shape) else 1
        N = U.shape[0]
        x, U = x.contiguous(), U.contiguous()
        out = torch.empty(shape[:axis] + [N] + shape[axis + 1:] + [2], device=x.device)
        L.check(L.load().dlwp_cmode_product(L.ptr(x), L.ptr(U), L.ptr(out), O, R, N, I, L.stream()))
        ctx.save_for_backward(x, U)
        ctx.dims = (O, R, N, I)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, U = ctx.saved_tensors
        O, R, N, I = ctx.dims
        gin, gU = torch.empty_like(x), torch.empty_like(U)
        g = gout.contiguous()          # kept alive until the launch is enqueued
        L.check(L.load().dlwp_cmode_product_bwd(L.ptr(x), L.ptr(U), L.ptr(g), L.ptr(gin), L.ptr(gU),
                                                O, R, N, I, L.stream()))
        return gin, gU, None


class TuckerSpectralWeight(nn.Module):
    """One layer's complex weight [Cin, Cout, m1, m2c] in Tucker form (parameters stored as real pairs [..., 2])."""

    def __init__(self, cin, cout, m1, m2c, rank, init_std):
        super().__init__()
        self.shape = (cin, cout, m1, m2c)
        self.rank = tucker_rank(self.shape, rank)
        # tltorch tucker init: every factor and the core N(0, s) with s chosen so that the reconstruction has std init_std.
        # The parameters are COMPLEX (stored as real pairs): a product of five complex numbers whose parts are N(0, s) has
        # second moment (2 s^2)^5, so that the reconstruction's real and imaginary parts each get std init_std -- the statistics
        # of this build's dense weights (fno_engine.FnoParamLayout.init: parts N(0, init_std)) -- with
        #   R (2 s^2)^5 = 2 init_std^2   <=>   s = (init_std / (4 sqrt(R)))^(1/5),   R = prod(rank).
        # (With parts N(0, s) and s = (init_std / sqrt(R))^(1/5), the real-tensor formula, the reconstruction comes out 5.7x too
        # large and a 4-layer network starts at a loss of ~175 instead of ~2.)
        r = math.sqrt(float(np.prod(self.rank)))
        s = (init_std / (4.0 * r)) ** (1.0 / (len(self.shape) + 1))
        self.core = nn.Parameter(torch.randn(*self.rank, 2) * s)
        self.factors = nn.ParameterList([nn.Parameter(torch.randn(d, rk, 2) * s) for d, rk in zip(self.shape, self.rank)])

    def dense(self):
        w = self.core
        for axis, U in enumerate(self.factors):
            w = _ModeProduct.apply(w, U, axis)
        return w                                        # [Cin, Cout, m1, m2c, 2]

    def dense_mode_major(self):
        return self.dense().permute(2, 3, 0, 1, 4).contiguous()   # [m1, m2c, Cin, Cout, 2]
