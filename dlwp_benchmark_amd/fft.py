"""rfft2 / irfft2 on libdlwpmi's LDS-staged FFT kernels (csrc/fft2d.hip), with their adjoints as autograd backward.

Reference call sites: torch.fft.rfft2 / irfft2(x, dim=(1, 2), norm="ortho") on channels-last [B, H, W, C] tensors in
AFNO2D.forward (src/nsbench/models/fourcastnet/fourcastnet.py:84,123; dlwpbench twin :85,124); rfftn / irfftn(norm="forward")
over the last dims of [B, C, H, W] in neuralop's SpectralConv.

Spectra are real tensors with a trailing (re, im) axis, [B, H, W//2+1, C, 2] (channels-last) or [B, C, H, W//2+1, 2]
(channels-first): gradients then follow the (dL/dRe, dL/dIm) convention of SURVEY.md App. D without detouring through
torch's complex autograd.  `torch.view_as_complex` gives torch.fft's layout when one is wanted.
"""
import ctypes as C

import torch

from . import lib as L

LAYOUTS = {"channels_last": 0, "channels_first": 1}
NORMS = {"backward": 0, None: 0, "ortho": 1, "forward": 2}
_PLANS = {}


def _plan(H, W):
    key = (int(H), int(W))
    if key not in _PLANS:
        h = C.c_void_p()
        L.check(L.load().dlwp_fft_plan_create(key[0], key[1], C.byref(h)))
        _PLANS[key] = h
    return _PLANS[key]


def _dims(x, layout):
    if layout == 0:
        B, H, W, Cc = x.shape
    else:
        B, Cc, H, W = x.shape
    return B, Cc, H, W


def _spec_shape(B, Cc, H, W, layout):
    return (B, H, W // 2 + 1, Cc, 2) if layout == 0 else (B, Cc, H, W // 2 + 1, 2)


def _run_r2c(x, layout, norm, adjoint):
    B, Cc, H, W = _dims(x, layout)
    X = torch.empty(_spec_shape(B, Cc, H, W, layout), device=x.device)
    L.check(L.load().dlwp_rfft2(_plan(H, W), L.ptr(x), L.ptr(X), B, Cc, layout, norm, adjoint, L.stream()))
    return X


def _run_c2r(X, W, layout, norm, adjoint):
    if layout == 0:
        B, H, _, Cc, _ = X.shape
        x = torch.empty(B, H, W, Cc, device=X.device)
    else:
        B, Cc, H, _, _ = X.shape
        x = torch.empty(B, Cc, H, W, device=X.device)
    work = torch.empty_like(X)
    L.check(L.load().dlwp_irfft2(_plan(H, W), L.ptr(X), L.ptr(x), L.ptr(work), B, Cc, layout, norm, adjoint, L.stream()))
    return x


class _RFFT2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, layout, norm):
        x = x.contiguous().float()
        ctx.cfg = (layout, norm, _dims(x, layout)[3])
        return _run_r2c(x, layout, norm, 0)

    @staticmethod
    def backward(ctx, gX):
        layout, norm, W = ctx.cfg
        return _run_c2r(gX.contiguous().float(), W, layout, norm, 1), None, None


class _IRFFT2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, W, layout, norm):
        ctx.cfg = (layout, norm)
        return _run_c2r(X.contiguous().float(), W, layout, norm, 0)

    @staticmethod
    def backward(ctx, gx):
        layout, norm = ctx.cfg
        return _run_r2c(gx.contiguous().float(), layout, norm, 1), None, None, None


def rfft2(x, layout="channels_last", norm="ortho"):
    """x [B, H, W, C] (channels_last) or [B, C, H, W] (channels_first) -> half spectrum with a trailing (re, im) axis."""
    return _RFFT2.apply(x, LAYOUTS[layout], NORMS[norm])


def irfft2(X, W, layout="channels_last", norm="ortho"):
    """Inverse of rfft2 for an output width W (the height is X's)."""
    return _IRFFT2.apply(X, int(W), LAYOUTS[layout], NORMS[norm])
