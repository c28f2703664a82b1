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


# ---- channels-last transforms with a planar WINDOW of the spectrum, [2 (re | im), B, r1 - r0, c1, C] (csrc/fft2d.hip,
# dlwp_rfft2_planar): the kept modes in the layout the AFNO mixer's block GEMMs read and write (afno_tiled.afno2d_fft)
def _run_r2c_planar(x, win, bs, norm, adjoint, mask=None, lam=0.0, out_bf16=False):
    """mask (X's layout and dtype) / lam: components of X are zeroed where |mask| <= lam (a soft-shrink derivative folded into the
    store).  out_bf16: the window is written as a bf16 array (bf16 storage of the AFNO block MLP's operands)."""
    B, H, W, Cc = x.shape
    r0, r1, c1 = win
    dt = torch.bfloat16 if out_bf16 else torch.float32
    X = torch.empty((B, r1 - r0, c1, Cc // bs, 2, bs) if bs else (2, B, r1 - r0, c1, Cc), device=x.device, dtype=dt)
    work = torch.empty(B, H, W // 2 + 1, Cc, 2, device=x.device)
    if mask is not None and (mask.numel() != X.numel() or mask.dtype != dt or not mask.is_contiguous()):
        raise L.DlwpError("rfft2_planar: the mask must be a contiguous tensor of the spectrum window's size and storage type")
    if mask is not None or out_bf16:
        L.check(L.load().dlwp_rfft2_planar_ex(_plan(H, W), L.ptr(x), L.ptr(X), L.ptr(work), L.ptr(mask) if mask is not None else None,
                                              float(lam), B, Cc, r0, r1, c1, bs, norm, adjoint, int(out_bf16), L.stream()))
        return X
    L.check(L.load().dlwp_rfft2_planar(_plan(H, W), L.ptr(x), L.ptr(X), L.ptr(work), B, Cc, r0, r1, c1, bs, norm, adjoint, L.stream()))
    return X


def _run_c2r_planar(X, H, W, win, bs, norm, adjoint, residual=None, residual2=None):
    B = X.shape[0] if bs else X.shape[1]
    Cc = X.shape[3] * bs if bs else X.shape[4]
    r0, r1, c1 = win
    x = torch.empty(B, H, W, Cc, device=X.device)
    work = torch.empty(B, H, W // 2 + 1, Cc, 2, device=X.device)
    for r in (residual, residual2):
        if r is not None and (tuple(r.shape) != (B, H, W, Cc) or r.dtype != torch.float32 or not r.is_contiguous()):
            raise L.DlwpError("irfft2_planar: a residual must be a contiguous fp32 field of the output's shape")
    if X.dtype == torch.bfloat16 or residual2 is not None:
        # (a bf16 spectrum window; both skips of an AFNO block -- the filter's own `+ x` and the block's outer one -- in the store)
        if residual2 is not None and residual is None:
            raise L.DlwpError("irfft2_planar: a second residual needs the first")
        L.check(L.load().dlwp_irfft2_planar_ex(_plan(H, W), L.ptr(X), L.ptr(x), L.ptr(work), L.ptr(residual) if residual is not None else None,
                                               L.ptr(residual2) if residual2 is not None else None, B, Cc, r0, r1, c1, bs, norm, adjoint,
                                               int(X.dtype == torch.bfloat16), L.stream()))
        return x
    L.check(L.load().dlwp_irfft2_planar(_plan(H, W), L.ptr(X), L.ptr(x), L.ptr(work), L.ptr(residual) if residual is not None else None,
                                        B, Cc, r0, r1, c1, bs, norm, adjoint, L.stream()))
    return x


class _RFFT2Planar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, win, bs, norm):
        x = x.contiguous().float()
        ctx.cfg = (win, bs, norm, x.shape[1], x.shape[2])
        return _run_r2c_planar(x, win, bs, norm, 0)

    @staticmethod
    def backward(ctx, gX):
        win, bs, norm, H, W = ctx.cfg
        return _run_c2r_planar(gX.contiguous().float(), H, W, win, bs, norm, 1), None, None, None


class _IRFFT2Planar(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, H, W, win, bs, norm):
        ctx.cfg = (win, bs, norm)
        return _run_c2r_planar(X.contiguous().float(), H, W, win, bs, norm, 0)

    @staticmethod
    def backward(ctx, gx):
        win, bs, norm = ctx.cfg
        return _run_r2c_planar(gx.contiguous().float(), win, bs, norm, 1), None, None, None, None, None


def _window(H, W, window):
    r0, r1, c1 = (0, H, W // 2 + 1) if window is None else (int(v) for v in window)
    if not (0 <= r0 < r1 <= H and 1 <= c1 <= W // 2 + 1):
        raise L.DlwpError(f"spectrum window rows [{r0}, {r1}) x {c1} columns outside {H} x {W // 2 + 1}")
    return r0, r1, c1


def rfft2_planar(x, norm="ortho", window=None, block=0):
    """x [B, H, W, C] -> [2, B, r1 - r0, c1, C]: plane 0 the real parts of torch.fft.rfft2(x, dim=(1, 2), norm=norm)[:, r0:r1, :c1],
    plane 1 the imaginary parts; window = (r0, r1, c1), default the whole half spectrum.  block = bs > 0 (a divisor of C): the
    block-planar layout [B, r1 - r0, c1, C // bs, 2, bs] instead (channel block, re | im, channel within the block)."""
    if block and x.shape[3] % block:
        raise L.DlwpError(f"rfft2_planar: channel block {block} does not divide C = {x.shape[3]}")
    return _RFFT2Planar.apply(x, _window(x.shape[1], x.shape[2], window), int(block), NORMS[norm])


def irfft2_planar(X, H, W, norm="ortho", window=None, block=0):
    """irfft2 (output H x W) of the spectrum that equals X inside the window and zero outside (X in rfft2_planar's layout)."""
    return _IRFFT2Planar.apply(X, int(H), int(W), _window(int(H), int(W), window), int(block), NORMS[norm])
