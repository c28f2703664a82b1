"""Window partition / reverse for shifted-window attention on libdlwpmi's gather kernels (csrc/window_ops.hip):
pad + roll + partition is ONE kernel, reverse + roll back + crop another (the reference runs each as a separate
full-tensor copy: nsbench swin_transformer.py:213-250, dlwpbench panguweather.py:283-317).
"""
import ctypes as C

import torch

from . import lib as L


class WindowSpec:
    """Geometry of one partition: token grid `dims` (3 axes, leading 1s for 2-D), window sizes, front / back pads,
    roll shift, per-axis padding mode ("constant" | "circular") and the order of the windows inside a sample
    (`order`: axes from slowest to fastest, e.g. (0, 1, 2) row-major for Swin, (2, 0, 1) longitude-major for Pangu)."""

    def __init__(self, dims, window, front=(0, 0, 0), back=(0, 0, 0), shift=(0, 0, 0), modes=("constant",) * 3, order=(0, 1, 2)):
        self.dims, self.window, self.front, self.shift = tuple(dims), tuple(window), tuple(front), tuple(shift)
        self.padded = tuple(f + d + b for f, d, b in zip(front, dims, back))
        assert all(p % w == 0 for p, w in zip(self.padded, window)), "padded grid must be a multiple of the window"
        self.nwin = tuple(p // w for p, w in zip(self.padded, window))
        self.nW = self.nwin[0] * self.nwin[1] * self.nwin[2]
        self.N = window[0] * window[1] * window[2]
        self.circ = tuple(int(m == "circular") for m in modes)
        stride, s = [0, 0, 0], 1
        for ax in reversed(order):
            stride[ax] = s
            s *= self.nwin[ax]
        self.wstride = tuple(stride)
        i3, l3 = C.c_int * 3, C.c_longlong * 3
        self._c = (i3(*self.dims), i3(*self.padded), i3(*self.front), i3(*self.window), l3(*self.wstride), i3(*self.circ))

    def c_args(self, shift):
        d, p, f, w, sw, circ = self._c
        return d, p, f, (C.c_int * 3)(*shift), w, sw, circ


def _gather(x, spec, shift, circ_override=None):
    B, Cc = x.shape[0], x.shape[-1]
    out = torch.empty(B * spec.nW, spec.N, Cc, device=x.device)
    d, p, f, s, w, sw, circ = spec.c_args(shift)
    if circ_override is not None:
        circ = (C.c_int * 3)(*circ_override)
    L.check(L.load().dlwp_window_gather(L.ptr(x), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ, L.stream()))
    return out


def _scatter(wins, spec, shift, B, sum_copies, residual=None):
    Cc = wins.shape[-1]
    out = torch.empty(B, spec.dims[0] * spec.dims[1] * spec.dims[2], Cc, device=wins.device)
    d, p, f, s, w, sw, circ = spec.c_args(shift)
    L.check(L.load().dlwp_window_scatter_add(L.ptr(wins), L.ptr(residual), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ,
                                             int(sum_copies), L.stream()))
    return out


class _PartitionFn(torch.autograd.Function):
    """tokens [B, L, C] -> windows [B * nW, N, C] of the padded, rolled canvas."""

    @staticmethod
    def forward(ctx, x, spec, shift):
        ctx.spec, ctx.shift, ctx.B = spec, shift, x.shape[0]
        return _gather(x.contiguous().float(), spec, shift)

    @staticmethod
    def backward(ctx, g):
        return _scatter(g.contiguous(), ctx.spec, ctx.shift, ctx.B, sum_copies=any(ctx.spec.circ)), None, None


class _PartitionFillFn(torch.autograd.Function):
    """partition of a tensor a token-wise Linear layer has already produced: padded positions hold `fill` (the layer's bias)
    instead of zero (dlwp_window_gather_fill); the fill's gradient is the column sum over the padded positions."""

    @staticmethod
    def forward(ctx, x, fill, spec, shift, c_lo=0):
        ctx.spec, ctx.shift, ctx.B, ctx.c_lo = spec, shift, x.shape[0], int(c_lo)
        # (axes with circular padding wrap -- their copies are copies of x rows and their gradients are summed back by the scatter;
        # the fill stands at the positions the CONSTANT axes pad)
        x = x.contiguous().float()
        B, Cc = x.shape[0], x.shape[-1]
        out = torch.empty(B * spec.nW, spec.N, Cc, device=x.device)
        d, p, f, s, w, sw, circ = spec.c_args(shift)
        fl = fill.detach().contiguous().float()
        L.check(L.load().dlwp_window_gather_fill(L.ptr(x), L.ptr(fl), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ, L.stream()))
        ctx.fill_slot = _fill_slot(fill)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        spec = ctx.spec
        gx = _scatter(g, spec, ctx.shift, ctx.B, sum_copies=any(spec.circ))
        gfill = ctx.fill_slot if ctx.fill_slot is not None else torch.zeros(g.shape[-1], device=g.device)
        d, p, f, s, w, sw, circ = spec.c_args(ctx.shift)
        L.check(L.load().dlwp_window_pad_colsum(L.ptr(g), L.ptr(gfill), ctx.B, g.shape[-1], d, p, f, s, w, sw, circ, ctx.c_lo,
                                                L.stream()))
        return gx, (None if ctx.fill_slot is not None else gfill), None, None, None


def _fill_slot(fill):
    """the parameter's slice of the flat gradient buffer (train_engine.flatten_parameters): kernels accumulate straight into it"""
    from .token_ops import _grad_slot
    return _grad_slot(fill)


class _ReverseFn(torch.autograd.Function):
    """windows [B * nW, N, C] -> tokens [B, L, C]: un-roll by `shift`, drop the padding (+ residual [B, L, C]: the block's skip
    connection, added by the same kernel; its gradient is the upstream gradient itself)."""

    @staticmethod
    def forward(ctx, wins, spec, shift, B, residual=None):
        ctx.spec, ctx.shift, ctx.has_res = spec, shift, residual is not None
        res = residual.reshape(B, -1, wins.shape[-1]).contiguous().float() if residual is not None else None
        return _scatter(wins.contiguous().float(), spec, shift, B, sum_copies=False, residual=res)

    @staticmethod
    def backward(ctx, g):
        # padded positions were dropped: they receive zero gradient whatever the padding mode of the forward partition
        return (_gather(g.contiguous(), ctx.spec, ctx.shift, circ_override=(0, 0, 0)), None, None, None,
                g if ctx.has_res else None)


def _identity(spec, shift):
    """One window = the whole unpadded, unshifted grid: tokens and windows are the same array."""
    return spec.nW == 1 and spec.padded == spec.dims and not any(s % p for s, p in zip(shift, spec.padded))


def position_maps(spec, fwd_shift, rev_shift, device):
    """int32 [nW, N] pair for dlwp_window_attn_bwd_tokens: (src_map, dst_map) = the token a window position was gathered from by
    partition(., spec, fwd_shift) and the token reverse(., spec, ., rev_shift) scatters it to; -1 = padding / cropped.  Built by
    running the gather kernels themselves over the token numbers (exact in fp32 below 2^24 tokens) and cached on the spec."""
    key = (tuple(fwd_shift), tuple(rev_shift), str(device))
    cache = spec.__dict__.setdefault("_position_maps", {})
    if key not in cache:
        assert not any(spec.circ), "position maps describe constant padding (every token in exactly one window position)"
        Ltok = spec.dims[0] * spec.dims[1] * spec.dims[2]
        assert Ltok < (1 << 24)
        idx = (torch.arange(Ltok, device=device, dtype=torch.float32) + 1.0).reshape(1, Ltok, 1).expand(1, Ltok, 4).contiguous()
        src = _gather(idx, spec, tuple(fwd_shift))[..., 0].round().to(torch.int32) - 1
        dst = _gather(idx, spec, tuple(rev_shift), circ_override=(0, 0, 0))[..., 0].round().to(torch.int32) - 1
        cache[key] = (src.reshape(spec.nW, spec.N).contiguous(), dst.reshape(spec.nW, spec.N).contiguous())
    return cache[key]


def partition(x, spec, shift=None, fill=None, fill_grad_from=0):
    """fill [C]: value of the padded positions (a bias), see _PartitionFillFn; None: zero.  fill_grad_from: channels below it are
    known to receive zero gradient at the padded positions (skipped by the fill's adjoint)."""
    shift = tuple(spec.shift if shift is None else shift)
    if _identity(spec, shift):
        return x.reshape(x.shape[0], spec.N, x.shape[-1])
    if fill is not None:
        return _PartitionFillFn.apply(x, fill, spec, shift, fill_grad_from)
    return _PartitionFn.apply(x, spec, shift)


def reverse(wins, spec, B, shift=None, residual=None):
    shift = tuple(spec.shift if shift is None else shift)
    if _identity(spec, shift):
        y = wins.reshape(B, spec.N, wins.shape[-1])
        if residual is None:
            return y
        from .token_ops import add_tokens
        return add_tokens(y, residual.reshape(y.shape))
    return _ReverseFn.apply(wins, spec, shift, B, residual)


class _PatchMergeFn(torch.autograd.Function):
    """tokens [B, H, W, C] -> [B, ceil(H/2), ceil(W/2), 4C] (dlwp_patch_merge: zero pad + the reference's four strided slices
    + cat, swin_transformer.py:291-312, as one gather kernel; the backward pass is its adjoint gather)."""

    @staticmethod
    def forward(ctx, x):
        B, H, W, Cc = x.shape
        x = x.contiguous().float()
        out = torch.empty(B, (H + 1) // 2, (W + 1) // 2, 4 * Cc, device=x.device)
        L.check(L.load().dlwp_patch_merge(L.ptr(x), L.ptr(out), B, H, W, Cc, 0, L.stream()))
        ctx.shape = (B, H, W, Cc)
        return out

    @staticmethod
    def backward(ctx, g):
        B, H, W, Cc = ctx.shape
        g = g.contiguous().float()
        gx = torch.empty(B, H, W, Cc, device=g.device)
        L.check(L.load().dlwp_patch_merge(L.ptr(g), L.ptr(gx), B, H, W, Cc, 1, L.stream()))
        return gx


def patch_merge(x):
    return _PatchMergeFn.apply(x)
