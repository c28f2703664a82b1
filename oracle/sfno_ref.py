"""TEST INFRASTRUCTURE ONLY — CPU oracle for the spherical (SFNO) path.  Never imported by the product; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

PARITY UNPINNED: the algorithm lives in the third-party package torch-harmonics (git 13aa492, pinned in the reference's
README.md:42-52, constructed at src/dlwpbench/models/fno/fno.py:183-200 and models/fourcastnet/fourcastnet.py:411-428),
which is absent from /root/reference and from this image.  This file restates its published algorithm (SURVEY.md
App. A-2): quadrature rules, the orthonormal Legendre recurrence, RealSHT / InverseRealSHT as rfft + dense Legendre
einsum, the "driscoll-healy" spectral layer and the SFNO network.  The transforms are anchored analytically
(tests/test_sht.py: agreement with scipy's spherical harmonics, exactness of the quadrature, round trips); the network
wiring has no fixture to be checked against and follows App. A-2's description:
    encoder (1x1 conv, act, 1x1 conv no bias) -> + pos_embed -> L x [norm0 -> SHT -> per-degree complex weight ->
    iSHT -> + inner skip (1x1 conv of the block input) -> act -> norm1 -> MLP (1x1 convs) -> + outer skip (identity)]
    -> concat network input (big_skip) -> decoder (1x1 conv, act, 1x1 conv no bias)
with GELU activations and "none" normalisation (configs/model/sfno.yaml:19).
The dlwpbench wrapper (SFNO2DModule.forward, fno.py:217-259) is restated in its working form (unet.py:64-111).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def legendre_gauss_weights(n, a=-1.0, b=1.0):
    xlg, wlg = np.polynomial.legendre.leggauss(n)
    return (b - a) * 0.5 * xlg + (b + a) * 0.5, wlg * (b - a) * 0.5


def clenshaw_curtiss_weights(n, a=-1.0, b=1.0):
    """Clenshaw-Curtis rule through the FFT of its cosine-series coefficients (Waldvogel's construction)."""
    assert n > 1
    tcc = np.cos(np.linspace(np.pi, 0, n))
    if n == 2:
        wcc = np.array([1.0, 1.0])
    else:
        n1 = n - 1
        N = np.arange(1, n1, 2)
        ln = len(N)
        m = n1 - ln
        v = np.concatenate([2 / N / (N - 2), 1 / N[-1:], np.zeros(m)])
        v = 0 - v[:-1] - v[-1:0:-1]
        g0 = -np.ones(n1)
        g0[ln] = g0[ln] + n1
        g0[m] = g0[m] + n1
        g = g0 / (n1 ** 2 - 1 + (n1 % 2))
        wcc = np.fft.ifft(v + g).real
        wcc = np.concatenate((wcc, wcc[:1]))
    return (b - a) * 0.5 * tcc + (b + a) * 0.5, wcc * (b - a) * 0.5


def legpoly(mmax, lmax, x, csphase=True):
    """[mmax, lmax, len(x)] orthonormal ("ortho") associated Legendre functions."""
    nmax = max(mmax, lmax)
    vdm = np.zeros((nmax, nmax, len(x)), dtype=np.float64)
    vdm[0, 0, :] = 1.0 / np.sqrt(4 * np.pi)
    for l in range(1, nmax):
        vdm[l - 1, l, :] = np.sqrt(2 * l + 1) * x * vdm[l - 1, l - 1, :]
        vdm[l, l, :] = np.sqrt((2 * l + 1) * (1 + x) * (1 - x) / 2 / l) * vdm[l - 1, l - 1, :]
    for l in range(2, nmax):
        for m in range(0, l - 1):
            vdm[m, l, :] = x * np.sqrt((2 * l - 1) / (l - m) * (2 * l + 1) / (l + m)) * vdm[m, l - 1, :] \
                - np.sqrt((l + m - 1) / (l - m) * (2 * l + 1) / (2 * l - 3) * (l - m - 1) / (l + m)) * vdm[m, l - 2, :]
    vdm = vdm[:mmax, :lmax]
    if csphase:
        for m in range(1, mmax, 2):
            vdm[m] *= -1
    return vdm


class SHT:
    """Tables of one (nlat, nlon, lmax, mmax, grid) transform pair."""

    def __init__(self, nlat, nlon, lmax=None, mmax=None, grid="legendre-gauss", dtype=torch.float32):
        if grid == "legendre-gauss":
            cost, w = legendre_gauss_weights(nlat)
        elif grid == "equiangular":
            cost, w = clenshaw_curtiss_weights(nlat)
        else:
            raise NotImplementedError(grid)
        self.nlat, self.nlon = nlat, nlon
        self.lmax, self.mmax = lmax or nlat, mmax or nlon // 2 + 1
        self.theta = np.flip(np.arccos(cost)).copy()
        pct = legpoly(self.mmax, self.lmax, np.cos(self.theta))
        self.pct = torch.from_numpy(pct).to(dtype)
        self.weights = torch.from_numpy(pct * w[None, None, :]).to(dtype)

    def forward(self, x):
        """x [..., nlat, nlon] real -> complex [..., lmax, mmax]."""
        X = 2.0 * math.pi * torch.fft.rfft(x, dim=-1, norm="forward")[..., :self.mmax]
        re = torch.einsum("...km,mlk->...lm", X.real, self.weights.to(x.dtype))
        im = torch.einsum("...km,mlk->...lm", X.imag, self.weights.to(x.dtype))
        return torch.complex(re, im)

    def inverse(self, X):
        """complex [..., lmax, mmax] -> real [..., nlat, nlon]."""
        re = torch.einsum("...lm,mlk->...km", X.real, self.pct.to(X.real.dtype))
        im = torch.einsum("...lm,mlk->...km", X.imag, self.pct.to(X.real.dtype))
        return torch.fft.irfft(torch.complex(re, im), n=self.nlon, dim=-1, norm="forward")


def conv1x1(x, p, name, bias=True):
    return F.conv2d(x, p[name + ".weight"], p[name + ".bias"] if bias else None)


def sfno_net(x, p, cfg):
    """x [B, in_chans, H, W] -> [B, out_chans, H, W]; parameter names follow dlwp_benchmark_amd/dlwpbench/sfno.py."""
    H, W, s = cfg["height"], cfg["width"], cfg.get("scale_factor", 1)
    h, w = H // s, W // s
    frac = cfg.get("hard_thresholding_fraction", 1.0)
    modes = min(int(h * frac), int(w // 2 * frac))
    down = SHT(H, W, modes, modes, cfg["grid"])
    inner = SHT(h, w, modes, modes, "legendre-gauss")
    n_layers = cfg["num_layers"]
    residual_in = x
    t = conv1x1(F.gelu(conv1x1(x, p, "encoder.0")), p, "encoder.2", bias=False)
    if "pos_embed" in p:
        t = t + p["pos_embed"]
    for i in range(n_layers):
        fwd_t = down if i == 0 else inner
        inv_t = down if i == n_layers - 1 else inner
        pre = f"blocks.{i}."
        if pre + "norm0.weight" in p:             # nn.InstanceNorm2d(embed_dim, eps=1e-6, affine=True) (App. A-2)
            t = F.instance_norm(t, weight=p[pre + "norm0.weight"], bias=p[pre + "norm0.bias"], eps=1e-6)
        res = t
        X = fwd_t.forward(t)
        if (fwd_t.nlat, fwd_t.nlon) != (inv_t.nlat, inv_t.nlon):
            res = inv_t.inverse(X)                                  # residual resampled to the output grid
        wc = torch.view_as_complex(p[pre + "filter.weight"].contiguous())
        Y = torch.einsum("bixy,iox->boxy", X, wc)
        t = inv_t.inverse(Y)
        if pre + "inner_skip.weight" in p:
            t = t + conv1x1(res, p, pre + "inner_skip")
        t = F.gelu(t)
        if pre + "norm1.weight" in p:
            t = F.instance_norm(t, weight=p[pre + "norm1.weight"], bias=p[pre + "norm1.bias"], eps=1e-6)
        if pre + "mlp.fc1.weight" in p:
            t = conv1x1(F.gelu(conv1x1(t, p, pre + "mlp.fc1")), p, pre + "mlp.fc2")
        t = t + res
    if cfg.get("big_skip", False):
        t = torch.cat([t, residual_in], dim=1)
    return conv1x1(F.gelu(conv1x1(t, p, "decoder.0")), p, "decoder.2", bias=False)


def sfno2d_rollout(constants, prescribed, prognostic, p, cfg):
    """SFNO2DModule.forward (fno.py:217-259) in its working form: out_t = prog_t[:, -1] + sfno(x_t)."""
    ctx, outs = cfg["context_size"], []
    for t in range(ctx, prognostic.shape[1]):
        if t == ctx:
            prog_t = prognostic[:, max(0, t - ctx):t]
        else:
            prog_t = torch.cat([prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        parts = [] if constants is None else [constants[:, 0]]
        if prescribed is not None:
            parts.append(prescribed[:, t - ctx:t].flatten(1, 2))
        parts.append(prog_t.flatten(1, 2))
        outs.append(prog_t[:, -1] + sfno_net(torch.cat(parts, dim=1), p, cfg))
    return torch.stack(outs, dim=1)
