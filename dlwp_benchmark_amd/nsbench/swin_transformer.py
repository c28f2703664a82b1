"""Drop-in counterpart of the reference's nsbench SwinTransformer (window-attention U-Net rollout).

Reference (file:line under /root/reference/src/nsbench/models/swintransformer/swin_transformer.py):
  WindowAttention :75-155, SwinTransformerBlock :158-258, PatchMerging :261-302, BasicLayer :305-408,
  PatchEmbed :411-452, SwinTransformer :455-706.  Constructor kwargs, forward signature and parameter /
  buffer names are kept so reference checkpoints load (`relative_position_index` is kept as a buffer for
  that reason only: the kernel derives the index from token coordinates).

The attention core softmax(scale q k^T + bias + mask) v runs as hand-written HIP kernels
(libdlwpmi dlwp_window_attn_fwd/bwd): scores never reach HBM, the shift mask is a per-window label
vector instead of an N x N tensor.  LayerNorm, qkv/proj/MLP/merging
Linear layers run on libdlwpmi's MFMA GEMM / LayerNorm kernels (token_ops.py).  Patch embedding, the U-decoder's
stride-2 transposed convolutions (GELU fused) and the 1x1 head are unfold / pixel-shuffle + the same GEMM.
Pad + roll + window partition (and the reverse) are one libdlwpmi gather kernel each (window_ops.py); stochastic depth
(drop_path_rate, shipped default 0.2) is a per-sample scale fused with the residual add (token_ops.DropPath).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib as L
from ..rollout_ops import ns_rollout
from .. import token_ops as _TO
from ..token_ops import _gemm_batched, _grad_slot, DropPath, DropPathPool, LayerNorm, Linear, Mlp, PatchConv2d, UpConvT2d, WgradBatch, norm_fork
from ..window_ops import WindowSpec, _gather, _scatter, partition, patch_merge, position_maps, reverse


class _WindowAttnFn(torch.autograd.Function):
    """softmax(scale q k^T + table[ia[q] + ib[k]][type] + mask) v on libdlwpmi: the fused flash-style kernels
    (dlwp_window_attn_fwd/bwd, head_dim <= 64) or, for the wide heads of the deep U-Net stages (small token maps), batched
    MFMA GEMMs around the row-softmax kernels (dlwp_window_softmax_fwd/bwd)."""

    @staticmethod
    def forward(ctx, qkv, table, ia, ib, labels, nW, heads, scale, qrange=None):
        lib = L.load()
        B_, N, C3 = qkv.shape
        ctx.qrange = (0, N) if qrange is None else (int(qrange[0]), int(qrange[1]))
        d = C3 // (3 * heads)
        table_param, table = table, table.contiguous()
        TB = table.shape[0]
        ntypes = table.shape[1] if table.dim() == 3 else 1
        # a bf16 qkv tensor (the projection wrote it as bf16: WindowAttention.forward asked for it) stays bf16 through the kernels,
        # and so do the output and both gradients (dlwp_window_attn_fwd_bf16 / _bwd_bf16)
        ctx.io_bf16 = (qkv.dtype == torch.bfloat16 and IO_BF16 and ctx.qrange == (0, N) and d <= 64
                       and lib.dlwp_window_attn_io_bf16_supported(N, d, TB, B_ * heads) == 1)
        qkv = qkv.contiguous() if ctx.io_bf16 else qkv.contiguous().float()
        # with a query range the kernels leave the rows outside it unwritten: they must read as zeros (the backward pass forms
        # g * o on every row, and 0 * uninitialised bits may be NaN)
        full = ctx.qrange == (0, N)
        out = (torch.empty if full else torch.zeros)(B_, N, heads * d, device=qkv.device, dtype=qkv.dtype)
        ctx.aux, ctx.cfg = (ia, ib, labels), (B_, nW, N, TB, ntypes, heads, d, scale)
        ctx.tslot = _grad_slot(table_param)
        ctx.wide = d > 64
        if ctx.wide:
            if N > 1024 or d % 4:
                raise L.DlwpError(f"window attention: head_dim {d} > 64 needs N <= 1024 tokens and head_dim % 4 == 0 (N = {N})")
            rs, hd = 3 * heads * d, heads * d
            p = torch.empty(B_, heads, N, N, device=qkv.device)
            _gemm_batched(qkv, qkv, p, N, N, d, rs, rs, N, 0, 1, B_, heads, (N * rs, d), (N * rs, d), (heads * N * N, N * N),
                          oB=hd)                                                   # s = q k^T
            L.check(lib.dlwp_window_softmax_fwd(L.ptr(p), L.ptr(table), L.ptr(ia), L.ptr(ib), L.ptr(labels), B_, nW, N, ntypes,
                                                heads, scale, L.stream()))
            _gemm_batched(p, qkv, out, N, d, N, N, rs, hd, 0, 0, B_, heads, (heads * N * N, N * N), (N * rs, d), (N * hd, d),
                          oB=2 * hd)                                               # o = p v
            ctx.save_for_backward(qkv, table, p)
            return out
        lse = (torch.empty if full else torch.zeros)(B_, heads, N, device=qkv.device)
        # earth-specific tables ([TB, types, heads]): one transposed copy per call, so that the workgroups read their
        # (type, head) slice contiguously instead of one cache line per entry
        packed = None
        if ntypes > 1:
            packed = torch.empty(ntypes * heads * TB, device=qkv.device)
            L.check(lib.dlwp_window_attn_pack_table(L.ptr(table), L.ptr(packed), TB, ntypes, heads, L.stream()))
        if ctx.io_bf16:
            L.check(lib.dlwp_window_attn_fwd_bf16(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels),
                                                  L.ptr(out), L.ptr(lse), B_, nW, N, TB, ntypes, heads, d, scale, L.stream()))
        else:
            L.check(lib.dlwp_window_attn_fwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels),
                                                    L.ptr(out), L.ptr(lse), B_, nW, N, TB, ntypes, heads, d, scale,
                                                    ctx.qrange[0], ctx.qrange[1], L.stream()))
        ctx.packed = packed
        ctx.save_for_backward(qkv, table, out, lse)
        return out

    @staticmethod
    def backward(ctx, gout):
        lib = L.load()
        ia, ib, labels = ctx.aux
        B_, nW, N, TB, ntypes, heads, d, scale = ctx.cfg
        g = gout.contiguous()          # kept alive until the launch is enqueued
        g = g.to(torch.bfloat16) if getattr(ctx, "io_bf16", False) else g.float()
        if ctx.wide:
            qkv, table, p = ctx.saved_tensors
            gtable = ctx.tslot if ctx.tslot is not None else torch.zeros_like(table)   # kernel accumulates
            rs, hd = 3 * heads * d, heads * d
            gqkv = torch.empty_like(qkv)
            sP, sQ, sO = (heads * N * N, N * N), (N * rs, d), (N * hd, d)
            _gemm_batched(p, g, gqkv, N, d, N, N, hd, rs, 1, 0, B_, heads, sP, sO, sQ, oC=2 * hd)     # dv = p^T do
            dp = torch.empty_like(p)
            _gemm_batched(g, qkv, dp, N, N, d, hd, rs, N, 0, 1, B_, heads, sO, sQ, sP, oB=2 * hd)     # dp = do v^T
            L.check(lib.dlwp_window_softmax_bwd(L.ptr(p), L.ptr(dp), L.ptr(gtable), L.ptr(ia), L.ptr(ib), B_, nW, N, ntypes,
                                                heads, scale, L.stream()))                            # dp <- scale ds
            _gemm_batched(dp, qkv, gqkv, N, d, N, N, rs, rs, 0, 0, B_, heads, sP, sQ, sQ, oB=hd)      # dq = ds k
            _gemm_batched(dp, qkv, gqkv, N, d, N, N, rs, rs, 1, 0, B_, heads, sP, sQ, sQ, oC=hd)      # dk = ds^T q
            return gqkv, (None if ctx.tslot is not None else gtable), None, None, None, None, None, None, None
        qkv, table, out, lse = ctx.saved_tensors
        gqkv = torch.empty_like(qkv)
        gtable = ctx.tslot if ctx.tslot is not None else torch.zeros_like(table)   # kernel accumulates
        if ctx.io_bf16:
            L.check(lib.dlwp_window_attn_bwd_bf16(L.ptr(qkv), L.ptr(table), L.ptr(ctx.packed), L.ptr(ia), L.ptr(ib), L.ptr(labels),
                                                  L.ptr(out), L.ptr(lse), L.ptr(g), L.ptr(gqkv), L.ptr(gtable), B_, nW, N, TB, ntypes,
                                                  heads, d, scale, L.stream()))
            return gqkv, (None if ctx.tslot is not None else gtable), None, None, None, None, None, None, None
        dsum = torch.empty_like(lse)
        slab = torch.empty(lib.dlwp_window_attn_bwd_slab_floats(B_, N, heads, TB), device=qkv.device)
        L.check(lib.dlwp_window_attn_bwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(ctx.packed), L.ptr(ia), L.ptr(ib), L.ptr(labels),
                                                L.ptr(out), L.ptr(lse), L.ptr(g), L.ptr(gqkv), L.ptr(gtable), L.ptr(dsum),
                                                L.ptr(slab), B_, nW, N, TB, ntypes, heads, d, scale, ctx.qrange[0], ctx.qrange[1],
                                                L.stream()))
        return gqkv, (None if ctx.tslot is not None else gtable), None, None, None, None, None, None, None


# opt-in (DLWP_SWIN_REAL_TOKENS=1): measured SLOWER on the Swin C4 step (6.92 vs 6.61 ms) -- with 5 % padding there are hardly any rows
# to save, and the gather then moves the 3C-wide projected tensor instead of the C-wide normalised one
REAL_TOKEN_FLOW = __import__("os").environ.get("DLWP_SWIN_REAL_TOKENS", "0") == "1"
FUSED_BWD = __import__("os").environ.get("DLWP_WINATTN_TOKENS", "1") != "0"      # env: A/B runs against the four-launch backward
FUSED_FWD = __import__("os").environ.get("DLWP_WINATTN_TOKENS_FWD", "1") != "0"  # env: ... and against gather + attention + scatter
IO_BF16 = __import__("os").environ.get("DLWP_WINATTN_IO_BF16", "1") != "0"       # env: ... and against fp32 qkv / output / gradient tensors


class _WindowAttnTokensFn(torch.autograd.Function):
    """reverse(crop) . attention . partition(pad with the qkv bias) of a block whose qkv projection ran on the real tokens, as ONE
    autograd node: forward = ONE launch on the token-layout qkv tensor where the wave-per-window family applies
    (dlwp_window_attn_fwd_tokens; otherwise dlwp_window_gather_fill, dlwp_window_attn_fwd_qrange, dlwp_window_scatter_add);
    backward = ONE launch (dlwp_window_attn_bwd_tokens) that reads the upstream gradient and writes the
    qkv gradient in the token layout through the two position maps and sums the padded positions' gradient into the bias's --
    instead of gather(gout) + attention backward + scatter(gqkv) + pad column sum (reference: EarthSpecificBlock.forward,
    src/dlwpbench/models/panguweather/panguweather.py:283-317)."""

    @staticmethod
    def applies(x, spec, d, table):
        """x: any CUDA tensor of the block (device check); d: head dimension"""
        return (FUSED_BWD and x.is_cuda and not any(spec.circ) and table.dim() == 3 and
                L.load().dlwp_window_attn_bwd_tokens_supported(spec.N, d, table.shape[0]) == 1)

    @staticmethod
    def wants_bf16_qkv(B, spec, heads, d):
        """the qkv projection may write bf16 (token_ops.Linear out_lowp): bf16 storage live and the one-launch forward applies"""
        from ..token_ops import _act_dtype
        return (FUSED_FWD and IO_BF16 and _act_dtype() == torch.bfloat16
                and L.load().dlwp_window_attn_fwd_tokens_supported(spec.N, d, B * spec.nW * heads) == 1)

    @staticmethod
    def forward(ctx, qkv_tok, fill, table, ia, ib, labels, spec, fwd_shift, rev_shift, heads, scale, qrange):
        lib = L.load()
        B, Ltok, C3 = qkv_tok.shape
        N, nW = spec.N, spec.nW
        d = C3 // (3 * heads)
        io = qkv_tok.dtype == torch.bfloat16          # bf16 tensors on all four sides (dlwp_window_attn_*_tokens io_bf16)
        x = qkv_tok.contiguous() if io else qkv_tok.contiguous().float()
        fl = fill.detach().contiguous().float()
        table_param, table = table, table.contiguous()
        TB, ntypes = table.shape[0], table.shape[1]
        qr = (0, N) if qrange is None else (int(qrange[0]), int(qrange[1]))
        full = qr == (0, N)
        packed = None
        if ntypes > 1:
            packed = torch.empty(ntypes * heads * TB, device=x.device)
            L.check(lib.dlwp_window_attn_pack_table(L.ptr(table), L.ptr(packed), TB, ntypes, heads, L.stream()))
        src_map, dst_map = position_maps(spec, fwd_shift, rev_shift, x.device)
        ctx.in_tokens = FUSED_FWD and lib.dlwp_window_attn_fwd_tokens_supported(N, d, B * nW * heads) == 1
        # (dlwp_window_attn_fwd_tokens writes zeros into the statistics of the rows it skips; the window-layout entry leaves them alone)
        lse = (torch.empty if (full or ctx.in_tokens) else torch.zeros)(B * nW, heads, N, device=x.device)
        ctx.io = io
        if io and not ctx.in_tokens:
            raise L.DlwpError("window attention: a bf16 qkv tensor needs the token-layout forward (dlwp_window_attn_fwd_tokens_supported)")
        if ctx.in_tokens:
            # ONE launch, no window-layout copy of qkv or of the output: the kernel reads token rows through src_map (padded
            # positions: the bias) and writes its rows to the tokens dst_map names (every token exactly once)
            y = torch.empty(B, Ltok, heads * d, device=x.device, dtype=torch.bfloat16 if io else torch.float32)
            L.check(lib.dlwp_window_attn_fwd_tokens(L.ptr(x), L.ptr(fl), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels),
                                                    L.ptr(src_map), L.ptr(dst_map), L.ptr(y), L.ptr(lse), B * nW, nW, N, Ltok, TB, ntypes,
                                                    heads, d, scale, qr[0], qr[1], int(io), L.stream()))
            ctx.save_for_backward(x, table, y, lse, fl)
        else:
            dd, pp, ff, ss, ww, sw, circ = spec.c_args(fwd_shift)
            qkv = torch.empty(B * nW, N, C3, device=x.device)
            L.check(lib.dlwp_window_gather_fill(L.ptr(x), L.ptr(fl), L.ptr(qkv), B, C3, dd, pp, ff, ss, ww, sw, circ, L.stream()))
            out = (torch.empty if full else torch.zeros)(B * nW, N, heads * d, device=x.device)
            L.check(lib.dlwp_window_attn_fwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out),
                                                    L.ptr(lse), B * nW, nW, N, TB, ntypes, heads, d, scale, qr[0], qr[1], L.stream()))
            y = _scatter(out, spec, rev_shift, B, sum_copies=False)
            ctx.save_for_backward(qkv, table, out, lse, fl)
        ctx.cfg = (B, Ltok, nW, N, TB, ntypes, heads, d, scale, qr)
        ctx.aux = (ia, ib, labels, packed, src_map, dst_map)
        ctx.tslot, ctx.fslot = _grad_slot(table_param), _grad_slot(fill)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        qkv, table, out, lse, fl = ctx.saved_tensors
        B, Ltok, nW, N, TB, ntypes, heads, d, scale, qr = ctx.cfg
        ia, ib, labels, packed, src_map, dst_map = ctx.aux
        g = gy.contiguous()
        if ctx.io:
            g = g if g.dtype == torch.bfloat16 else g.to(torch.bfloat16)
        else:
            g = g.float()
        gqkv = torch.empty(B, Ltok, 3 * heads * d, device=g.device, dtype=torch.bfloat16 if ctx.io else torch.float32)
        gtable = ctx.tslot if ctx.tslot is not None else torch.zeros_like(table)          # the kernel accumulates into both
        gfill = ctx.fslot if ctx.fslot is not None else torch.zeros(3 * heads * d, device=g.device)
        L.check(lib.dlwp_window_attn_bwd_tokens(L.ptr(qkv), L.ptr(fl) if ctx.in_tokens else None, L.ptr(table), L.ptr(packed), L.ptr(ia),
                                                L.ptr(ib), L.ptr(labels), L.ptr(out), L.ptr(lse), L.ptr(g), L.ptr(dst_map), L.ptr(src_map),
                                                L.ptr(gqkv), L.ptr(gfill), L.ptr(gtable), B * nW, nW, N, Ltok, TB, ntypes, heads, d, scale,
                                                qr[0], qr[1], int(ctx.io), L.stream()))
        return (gqkv, None if ctx.fslot is not None else gfill, None if ctx.tslot is not None else gtable) + (None,) * 9


def window_attention_tokens(qkv_tok, fill, table, ia, ib, labels, spec, fwd_shift, rev_shift, heads, scale, qrange=None):
    """tokens [B, L, 3C] (a qkv projection of the real tokens) -> attention output in the token layout [B, L, C]"""
    return _WindowAttnTokensFn.apply(qkv_tok, fill, table, ia, ib, labels, spec, tuple(fwd_shift), tuple(rev_shift), heads, scale, qrange)


def window_attention_core(qkv, table, ia, ib, labels, nW, heads, scale, qrange=None):
    """qrange (lo, hi): window tokens outside it are padding the caller crops afterwards -- still keys / values, but their own
    rows may be left uncomputed (dlwp_window_attn_fwd_qrange)."""
    return _WindowAttnFn.apply(qkv, table, ia, ib, labels, nW, heads, scale, qrange)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


def _tokens_to_windows(x, ws):
    B, H, W, C = x.shape
    wh, ww = ws
    return x.view(B, H // wh, wh, W // ww, ww, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, wh * ww, C)


def _windows_to_tokens(wins, ws, H, W):
    wh, ww = ws
    B = wins.shape[0] // ((H // wh) * (W // ww))
    return wins.view(B, H // wh, W // ww, wh, ww, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def _pad_hw(t, pad_h, pad_w, modes, dims_last=False):
    """Pad the H (bottom) and W (right) axes of [B,H,W,C] (or [B,C,H,W] with dims_last) with per-axis modes."""
    mode_h, mode_w = modes
    if not (pad_h or pad_w):
        return t
    if mode_h == mode_w:
        return F.pad(t, (0, pad_w, 0, pad_h) if dims_last else (0, 0, 0, pad_w, 0, pad_h), mode=mode_h)
    if pad_w:
        t = F.pad(t, (0, pad_w) if dims_last else (0, 0, 0, pad_w), mode=mode_w)
    if pad_h:
        t = F.pad(t, (0, 0, 0, pad_h) if dims_last else (0, 0, 0, 0, 0, pad_h), mode=mode_h)
    return t


class WindowAttention(nn.Module):
    def __init__(self, dim, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0., proj_drop=0.):
        super().__init__()
        if attn_drop or proj_drop:
            raise NotImplementedError("dropout is not on the MI355X hot path (configs use 0.0)")
        self.dim, self.window_size, self.num_heads = dim, _pair(window_size), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        Wh, Ww = self.window_size
        self.relative_position_bias_table = nn.Parameter(torch.zeros((2 * Wh - 1) * (2 * Ww - 1), num_heads))
        ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")
        ys, xs = ys.reshape(-1), xs.reshape(-1)
        index = (ys[:, None] - ys[None, :] + Wh - 1) * (2 * Ww - 1) + (xs[:, None] - xs[None, :] + Ww - 1)
        self.register_buffer("relative_position_index", index)   # checkpoint compatibility only
        # the index is additive in query and key: index[q][k] = ia[q] + ib[k]
        self.register_buffer("_ia", (ys * (2 * Ww - 1) + xs).to(torch.int32), persistent=False)
        self.register_buffer("_ib", ((Wh - 1 - ys) * (2 * Ww - 1) + (Ww - 1 - xs)).to(torch.int32), persistent=False)
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        nn.init.trunc_normal_(self.relative_position_bias_table, std=.02)

    def forward(self, x, labels=None, nW=1, wbatch=None, out_lowp=False):
        """x [nW*B, N, C]; labels: int32 [nW, N] region labels of the shift mask (None: no mask); wbatch: the block's
        token_ops.WgradBatch (the weight gradients of qkv and proj join the block's one launch); out_lowp: the projection is
        written as bf16 where its operands are (bf16 storage) for a consumer that reads bf16 windows (window_ops.reverse)."""
        # bf16 storage: where the kernels take bf16 windows (dlwp_window_attn_io_bf16_supported) the projection writes qkv as bf16
        # and the attention output, its gradient and the qkv gradient are bf16 arrays as well: no cast launch on either side
        N, C = x.shape[-2], x.shape[-1]
        lowp_qkv = (IO_BF16 and x.is_cuda and x.dtype == torch.bfloat16 and _TO._act_dtype() == torch.bfloat16 and (C // self.num_heads) <= 64
                    and L.load().dlwp_window_attn_io_bf16_supported(N, C // self.num_heads, self.relative_position_bias_table.shape[0],
                                                                    x.shape[0] * self.num_heads) == 1)
        y = _WindowAttnFn.apply(self.qkv(x, wbatch=wbatch, out_lowp=lowp_qkv), self.relative_position_bias_table, self._ia, self._ib, labels,
                                nW, self.num_heads, float(self.scale))
        return self.proj(y, wbatch=wbatch, out_lowp=out_lowp)

    def core(self, qkv_windows, labels=None, nW=1):
        """attention on windows of an already projected qkv tensor [nW*B, N, 3C] (SwinTransformerBlock's real-token flow)"""
        return _WindowAttnFn.apply(qkv_windows, self.relative_position_bias_table, self._ia, self._ib, labels, nW, self.num_heads,
                                   float(self.scale))


class SwinTransformerBlock(nn.Module):
    wgrad_batch_block = True      # the block's weight-gradient writes land together at its first layer's backward (token_ops.WgradBatch)

    def __init__(self, dim, num_heads, window_size=7, shift_size=0, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm,
                 padding_mode: str = "constant"):
        super().__init__()
        # window / shift are (h, w) pairs; an int means a square window (nsbench).  padding_mode: one mode or a
        # (mode_h, mode_w) pair (dlwpbench: constant latitude, circular longitude)
        window_size, shift_size = _pair(window_size), _pair(shift_size)
        assert all(0 <= s < w for s, w in zip(shift_size, window_size)), "shift_size must in 0-window_size"
        self.window_size, self.shift_size, self.padding_mode = window_size, shift_size, _pair(padding_mode)
        self.norm1 = norm_layer(dim)
        self.attn = WindowAttention(dim, _pair(window_size), num_heads, qkv_bias, qk_scale, attn_drop, drop)
        # reference :193: DropPath(drop_path) if drop_path > 0. else nn.Identity() -- no parameters either way
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.H = self.W = None
        # The reference pads / rolls / partitions norm1(x) and runs qkv and proj on every WINDOW token (:229-250).  Both are token-wise
        # Linear layers and partition is a gather, so they commute with it exactly: qkv on the real tokens (a GEMM input straight from
        # the LayerNorm: bf16 under bf16 storage), the projected tensor partitioned -- circular copies are copies of the projected rows,
        # constant pads hold Linear(0) = the qkv bias (dlwp_window_gather_fill) -- attention, reverse, proj on the real tokens with the
        # residual in its epilogue.  Same result (tests/test_gpu_swin.py); it pays where much of a window is padding (Pangu: half), not
        # here -- see REAL_TOKEN_FLOW.
        self.real_token_flow = REAL_TOKEN_FLOW

    def forward(self, x, labels):
        B, L_, C = x.shape
        H, W, ws, sh = self.H, self.W, self.window_size, self.shift_size
        assert L_ == H * W, "input feature has wrong size"
        shifted = sh[0] > 0 or sh[1] > 0
        if C % 4 == 0:
            # pad + roll + partition and reverse + roll back + crop are one gather kernel each (window_ops.py)
            # the skip connections leave the LayerNorm nodes (norm_fork) so that their gradients join the LayerNorm backward
            # kernels; the first residual add rides the reverse kernel, the second fc2's epilogue
            spec = self._spec(H, W)
            wb = WgradBatch()        # this application's four weight gradients (qkv, proj, fc1, fc2) in one launch
            if self.real_token_flow:
                skip, t = norm_fork(self.norm1, x, gemm_input=True)
                const_pad = any((f or b) and not c for f, b, c in zip(spec.front, [p - f - d for p, f, d in zip(spec.padded, spec.front, spec.dims)], spec.circ))
                qkv = partition(self.attn.qkv(t, wbatch=wb), spec, fill=self.attn.qkv.bias if const_pad else None)
                t = self.attn.core(qkv, labels if shifted else None, spec.nW)
                if self.drop_path.active:
                    skip, t = norm_fork(self.norm2, self.drop_path.branch(self.attn.proj, reverse(t, spec, B), skip, wbatch=wb), gemm_input=True)
                    return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)
                skip, t = norm_fork(self.norm2, self.attn.proj(reverse(t, spec, B), residual=skip, wbatch=wb), gemm_input=True)
                return self.mlp(t, residual=skip, wbatch=wb)
            # (bf16 storage: norm1 writes bf16 rows, the gather moves them as they are and qkv reads them; proj writes bf16 windows
            # for the scatter, whose adjoint hands proj's backward products a bf16 gradient)
            skip, t = norm_fork(self.norm1, x, gemm_input=True)
            t = self.attn(partition(t, spec), labels if shifted else None, spec.nW, wbatch=wb, out_lowp=True)
            if self.drop_path.active:        # training with stochastic depth (:255-256): per-sample scale + residual add
                if _TO.DROPPATH_FUSED:       # ... inside the scatter kernel
                    mask = self.drop_path.mask(B, t.device)
                    skip, t = norm_fork(self.norm2, reverse(t, spec, B, residual=skip, row_scale=mask), gemm_input=True)
                else:
                    skip, t = norm_fork(self.norm2, self.drop_path(reverse(t, spec, B), residual=skip), gemm_input=True)
                return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)
            skip, t = norm_fork(self.norm2, reverse(t, spec, B, residual=skip), gemm_input=True)
            return self.mlp(t, residual=skip, wbatch=wb)
        t = self.norm1(x).view(B, H, W, C)
        t = _pad_hw(t, (ws[0] - H % ws[0]) % ws[0], (ws[1] - W % ws[1]) % ws[1], self.padding_mode)
        Hp, Wp = t.shape[1], t.shape[2]
        if shifted:
            t = torch.roll(t, shifts=(-sh[0], -sh[1]), dims=(1, 2))
        nW = (Hp // ws[0]) * (Wp // ws[1])
        t = self.attn(_tokens_to_windows(t, ws), labels if shifted else None, nW)
        t = _windows_to_tokens(t, ws, Hp, Wp)
        if shifted:
            t = torch.roll(t, shifts=(sh[0], sh[1]), dims=(1, 2))
        t = t[:, :H, :W, :].reshape(B, H * W, C)
        if self.drop_path.active:
            x = self.drop_path(t, residual=x)
            return self.drop_path.branch(self.mlp, self.norm2(x), x)
        x = x + t
        return self.mlp(self.norm2(x), residual=x)   # residual add fused into fc2's epilogue

    def _spec(self, H, W):
        key = (H, W)
        if getattr(self, "_spec_key", None) != key:
            ws, sh = self.window_size, self.shift_size
            self._spec_val = WindowSpec((1, H, W), (1, ws[0], ws[1]), back=(0, (ws[0] - H % ws[0]) % ws[0], (ws[1] - W % ws[1]) % ws[1]),
                                        shift=(0, sh[0], sh[1]), modes=("constant", self.padding_mode[0], self.padding_mode[1]))
            self._spec_key = key
        return self._spec_val


class PatchMerging(nn.Module):
    def __init__(self, dim, norm_layer=LayerNorm, padding_mode: str = "constant"):
        super().__init__()
        self.padding_mode = _pair(padding_mode)
        self.reduction = Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x, H, W):
        B, L_, C = x.shape
        assert L_ == H * W, "input feature has wrong size"
        x = x.view(B, H, W, C)
        zero_pad = (not H % 2 or self.padding_mode[0] == "constant") and (not W % 2 or self.padding_mode[1] == "constant")
        if zero_pad:      # pad + four strided slices + cat as one gather kernel
            x = patch_merge(x)
        else:
            x = _pad_hw(x, H % 2, W % 2, self.padding_mode)
            x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)
        return self.reduction(self.norm(x.reshape(B, -1, 4 * C)))


class BasicLayer(nn.Module):
    def __init__(self, dim, depth, num_heads, window_size=7, mlp_ratio=4., qkv_bias=True, qk_scale=None, drop=0.,
                 attn_drop=0., drop_path=0., norm_layer=LayerNorm, downsample=None, use_checkpoint=False,
                 padding_mode: str = "constant"):
        super().__init__()
        window_size = tuple(int(v) for v in _pair(window_size))
        self.window_size, self.shift_size, self.depth = window_size, (window_size[0] // 2, window_size[1] // 2), depth
        self.blocks = nn.ModuleList([
            SwinTransformerBlock(dim, num_heads, window_size, (0, 0) if i % 2 == 0 else self.shift_size, mlp_ratio,
                                 qkv_bias, qk_scale, drop, attn_drop,
                                 drop_path[i] if isinstance(drop_path, list) else drop_path,
                                 norm_layer=norm_layer, padding_mode=padding_mode) for i in range(depth)])
        self.downsample = downsample(dim=dim, norm_layer=norm_layer, padding_mode=padding_mode) if downsample else None
        self._labels = {}

    def shift_labels(self, Hp, Wp, device):
        """Region labels of the cyclically shifted canvas, one int32 vector per window (reference :377-393
        builds the same regions and expands them to an [nW,N,N] -100/0 mask on every forward)."""
        key = (Hp, Wp, str(device))
        if key not in self._labels:
            ws, sh = self.window_size, self.shift_size

            def axis_labels(n, w, s):
                lab = torch.zeros(n, dtype=torch.int32)
                lab[n - w:n - s] = 1
                if s > 0:
                    lab[n - s:] = 2
                return lab
            img = axis_labels(Hp, ws[0], sh[0])[:, None] * 3 + axis_labels(Wp, ws[1], sh[1])[None, :]
            lab = img.view(Hp // ws[0], ws[0], Wp // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(-1, ws[0] * ws[1])
            self._labels[key] = lab.contiguous().to(device)
        return self._labels[key]

    def forward(self, x, H, W):
        ws = self.window_size
        labels = self.shift_labels(math.ceil(H / ws[0]) * ws[0], math.ceil(W / ws[1]) * ws[1], x.device)
        for blk in self.blocks:
            blk.H, blk.W = H, W
            x = blk(x, labels)
        if self.downsample is not None:
            return x, H, W, self.downsample(x, H, W), (H + 1) // 2, (W + 1) // 2
        return x, H, W, x, H, W


class PatchEmbed(nn.Module):
    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None, padding_mode: str = "constant"):
        super().__init__()
        self.patch_size, self.embed_dim, self.padding_mode = _pair(patch_size), embed_dim, _pair(padding_mode)
        self.proj = PatchConv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        _, _, H, W = x.shape
        ph, pw = self.patch_size
        x = _pad_hw(x, (ph - H % ph) % ph, (pw - W % pw) % pw, self.padding_mode, dims_last=True)
        x = self.proj(x)
        if self.norm is not None:
            Wh, Ww = x.shape[2], x.shape[3]
            x = self.norm(x.flatten(2).transpose(1, 2)).transpose(1, 2).reshape(-1, self.embed_dim, Wh, Ww)
        return x


def absolute_position_tokens(embed, Wh, Ww):
    """`ape=True` (reference nsbench :640-643, dlwpbench :650-653): the [1, E, Wh0, Ww0] embedding as [1, Wh*Ww, E] tokens,
    resized bicubically when the token grid is not the one it was built for (on the shipped grids the resize is the identity)."""
    if embed.shape[2] != Wh or embed.shape[3] != Ww:
        embed = F.interpolate(embed, size=(Wh, Ww), mode="bicubic")
    return embed.flatten(2).transpose(1, 2)


_NORMS = {"nn.LayerNorm": LayerNorm, "th.nn.LayerNorm": LayerNorm, "torch.nn.LayerNorm": LayerNorm}


class SwinTransformer(nn.Module):
    def __init__(self, context_size: int = 10, pretrain_img_size=224, patch_size=4, in_chans=3, out_chans=1,
                 embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24], mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.2, norm_layer="nn.LayerNorm",
                 ape=False, patch_norm=True, frozen_stages=-1, use_checkpoint=False, padding_mode: str = "constant",
                 window_size=None, **kwargs):
        """window_size (extra kwarg, not in the reference): None = the reference behaviour (every stage attends over its
        whole feature map, :528); an int gives classic Swin windows (e.g. 7 for BASELINE C4) with padding to multiples."""
        super().__init__()
        if frozen_stages >= 0:
            raise NotImplementedError("frozen_stages >= 0 (a fine-tuning option: stop gradients of the first stages) is not "
                                      "built; the shipped configs use -1")
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("dropout is not on the MI355X hot path (the shipped configs use drop_rate 0 and "
                                      "attn_drop_rate 0)")
        norm = _NORMS[norm_layer] if isinstance(norm_layer, str) else norm_layer   # registry instead of eval()
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]    # stochastic depth decay rule (:542)
        self.context_size, self.num_layers, self.embed_dim = context_size, len(depths), embed_dim
        self.patch_embed = PatchEmbed(patch_size, in_chans * context_size, embed_dim, norm if patch_norm else None,
                                      padding_mode)
        resolution = pretrain_img_size // patch_size
        self.ape = ape
        if ape:     # learned [1, E, Wh0, Ww0] embedding added to the embedded patches (reference :530-537)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, resolution, resolution))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            # window = the stage's whole feature map (reference :528): global attention with a half-map shift
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i],
                                          resolution if window_size is None else window_size, mlp_ratio,
                                          qkv_bias, qk_scale, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])],
                                          norm_layer=norm,
                                          downsample=PatchMerging if i < self.num_layers - 1 else None,
                                          padding_mode=padding_mode))
            resolution //= 2
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i, nf in enumerate(self.num_features):
            self.add_module(f"norm{i}", norm(nf))
        self.decoder = nn.ModuleList()
        for idx, i in enumerate(reversed(range(self.num_layers))):
            ch = int(embed_dim * 2 ** i)
            # nn.Sequential(ConvTranspose2d, GELU) keeps the reference's parameter names `decoder.{idx}.0.*`;
            # the GELU is applied inside the transposed convolution's GEMM epilogue (one_step)
            self.decoder.append(nn.Sequential(
                UpConvT2d(ch if idx == 0 else 2 * ch, ch if i == 0 else ch // 2, kernel_size=2, stride=2),
                nn.GELU()))
        self.final = PatchConv2d(embed_dim, out_chans, kernel_size=1)

    def one_step(self, x):
        if getattr(self, "_drop_pool", None) is None:      # built lazily: after construction, copies and loads
            object.__setattr__(self, "_drop_pool", DropPathPool(self))
        self._drop_pool.draw(x.shape[0], x.device)        # every block's stochastic-depth mask for this call, one draw
        x = self.patch_embed(x)
        Wh, Ww = x.shape[2], x.shape[3]
        x = x.flatten(2).transpose(1, 2)
        if self.ape:
            x = x + absolute_position_tokens(self.absolute_pos_embed, Wh, Ww).to(x.dtype)
        # U-decoder on channels-last tokens (reference :580-591 / one_step): stage outputs stay [B, H, W, C], the transposed
        # convolutions are a GEMM + one interleave kernel each, the 1 x 1 head is a GEMM; NCHW only for the returned frame
        feats = []
        for i, layer in enumerate(self.layers):
            x_out, H, W, x, Wh, Ww = layer(x, Wh, Ww)
            x_out = getattr(self, f"norm{i}")(x_out)
            feats.append(x_out.reshape(-1, H, W, self.num_features[i]))
        feats.reverse()
        y = None
        for idx, up in enumerate(self.decoder):
            y = up[0].forward_tokens(feats[idx] if idx == 0 else torch.cat([feats[idx], y], dim=-1), act=1)   # GELU fused
        y = self.final.forward_tokens(y)                      # [B, H, W, out]
        return y.permute(0, 3, 1, 2)

    def forward(self, x: torch.Tensor, teacher_forcing_steps: int = 50) -> torch.Tensor:
        # reference :597-640; the sliding window is advanced by one kernel per lead time (rollout_ops.advance)
        return ns_rollout(self.one_step, x, teacher_forcing_steps, self.context_size)
