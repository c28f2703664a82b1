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


_BF = torch.bfloat16


def _gather(x, spec, shift, circ_override=None, scale=None, out_dtype=torch.float32):
    """x (fp32 or bf16) -> windows of `out_dtype`; scale [B] fp32: sample b's values times scale[b]."""
    B, Cc = x.shape[0], x.shape[-1]
    out = torch.empty(B * spec.nW, spec.N, Cc, device=x.device, dtype=out_dtype)
    d, p, f, s, w, sw, circ = spec.c_args(shift)
    if circ_override is not None:
        circ = (C.c_int * 3)(*circ_override)
    if x.dtype == _BF or out_dtype == _BF or scale is not None:
        L.check(L.load().dlwp_window_gather_ex(L.ptr(x), None, L.ptr(scale), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ,
                                               int(x.dtype == _BF) | (2 if out_dtype == _BF else 0), L.stream()))
        return out
    L.check(L.load().dlwp_window_gather(L.ptr(x), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ, L.stream()))
    return out


def _scatter(wins, spec, shift, B, sum_copies, residual=None, scale=None, out_dtype=torch.float32):
    """windows (fp32 or bf16) -> tokens of `out_dtype`: residual (fp32) + scale[b] * scatter."""
    Cc = wins.shape[-1]
    out = torch.empty(B, spec.dims[0] * spec.dims[1] * spec.dims[2], Cc, device=wins.device, dtype=out_dtype)
    d, p, f, s, w, sw, circ = spec.c_args(shift)
    if wins.dtype == _BF or out_dtype == _BF or scale is not None:
        L.check(L.load().dlwp_window_scatter_ex(L.ptr(wins), L.ptr(residual), L.ptr(scale), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ,
                                                int(sum_copies), int(wins.dtype == _BF) | (2 if out_dtype == _BF else 0), L.stream()))
        return out
    L.check(L.load().dlwp_window_scatter_add(L.ptr(wins), L.ptr(residual), L.ptr(out), B, Cc, d, p, f, s, w, sw, circ,
                                             int(sum_copies), L.stream()))
    return out


def _keep(t):
    """contiguous, fp32 unless it already is a bf16 array (bf16 storage: LayerNorm outputs, low-precision Linear outputs)"""
    t = t.contiguous()
    return t if t.dtype == _BF else t.float()


class _PartitionFn(torch.autograd.Function):
    """tokens [B, L, C] -> windows [B * nW, N, C] of the padded, rolled canvas."""

    @staticmethod
    def forward(ctx, x, spec, shift):
        ctx.spec, ctx.shift, ctx.B = spec, shift, x.shape[0]
        x = _keep(x)               # a bf16 input (a LayerNorm output written for the GEMM that follows) stays bf16 through the gather
        ctx.dtype = x.dtype
        return _gather(x, spec, shift, out_dtype=x.dtype)

    @staticmethod
    def backward(ctx, g):
        return _scatter(_keep(g), ctx.spec, ctx.shift, ctx.B, sum_copies=any(ctx.spec.circ), out_dtype=ctx.dtype), None, None


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
    def forward(ctx, wins, spec, shift, B, residual=None, row_scale=None):
        """row_scale [B] fp32: tokens = residual + row_scale[b] * reverse(wins)[b] (stochastic depth of the attention branch in the
        same pass); wins may be a bf16 array (a projection written with out_lowp), the tokens are fp32."""
        ctx.spec, ctx.shift, ctx.has_res, ctx.row_scale = spec, shift, residual is not None, row_scale
        res = residual.reshape(B, -1, wins.shape[-1]).contiguous().float() if residual is not None else None
        wins = _keep(wins)
        ctx.dtype = wins.dtype
        return _scatter(wins, spec, shift, B, sum_copies=False, residual=res, scale=row_scale)

    @staticmethod
    def backward(ctx, g):
        # padded positions were dropped: they receive zero gradient whatever the padding mode of the forward partition
        return (_gather(g.contiguous().float(), ctx.spec, ctx.shift, circ_override=(0, 0, 0), scale=ctx.row_scale, out_dtype=ctx.dtype),
                None, None, None, g if ctx.has_res else None, None)


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


def reverse(wins, spec, B, shift=None, residual=None, row_scale=None):
    shift = tuple(spec.shift if shift is None else shift)
    if _identity(spec, shift):
        y = wins.reshape(B, spec.N, wins.shape[-1])
        if row_scale is not None:
            from .token_ops import _ScaleRowsAddFn
            return _ScaleRowsAddFn.apply(y, row_scale, residual.reshape(y.shape) if residual is not None else None)
        if residual is None:
            return y
        from .token_ops import add_tokens
        return add_tokens(y, residual.reshape(y.shape))
    return _ReverseFn.apply(wins, spec, shift, B, residual, row_scale)


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
