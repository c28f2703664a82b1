"""AFNO2D for grids of any size, as strided-batched MFMA GEMMs (the LDS-resident kernel of csrc/afno.hip covers small
token grids with block size <= 16; everything else -- e.g. the FourCastNet-scale 90x180 grid with 768 channels, or
720x1440 at patch 1 -- runs here).

Reference: AFNO2D.forward, src/nsbench/models/fourcastnet/fourcastnet.py:77-126 (dlwpbench twin :78-127).  Same
arithmetic as the fused kernel, x [B, H, W, C] channels-last:
    1. W-axis real DFT ("ortho"), only the c1 kept columns:        T1[b,h][(re|im, kw)][c]  = F1 . x[b,h]          batch (b,h)
    2. H-axis complex DFT, only the kept rows [r0, r1):             X[re|im][b][kh][kw,c]    = E2 . T1[b][(h, re|im)]   batch (b, plane)
    3. block-diagonal complex MLP per mode (ReLU, then soft-shrink): two layers of batched GEMMs over the channel blocks on
       the real image [[Wr, Wi], [-Wi, Wr]] of the weights (csrc/afno.hip: dlwp_afno_wq_expand)
    4. inverse H-axis DFT (kept rows -> all rows), 5. inverse W-axis real DFT + the residual `+ x` in the GEMM epilogue.
Modes outside the kept window are zero in the reference (:98-117) and simply never computed here.
The kept window follows the reference's quirk: it is computed from H only (:92-93).
"""
import math

import numpy as np
import torch

from . import lib as L
from .sht import _TableGemm
from .token_ops import _act_dtype, _gemm_batched, _grad_slot

_TABLES = {}


def kept_window(H, W, frac):
    total = H // 2 + 1
    kept = int(total * frac)
    return max(0, total - kept), min(H, total + kept), min(W // 2 + 1, kept)


def _tables(H, W, frac, device):
    key = (H, W, float(frac), str(device))
    if key in _TABLES:
        return _TABLES[key]
    r0, r1, c1 = kept_window(H, W, frac)
    R = r1 - r0
    kw, w = np.arange(c1), np.arange(W)
    aw = 2.0 * np.pi * np.outer(kw, w) / W                       # [c1, W]
    F1 = np.concatenate([np.cos(aw), -np.sin(aw)], axis=0) / math.sqrt(W)             # rows (re|im, kw)
    ck = np.where((kw == 0) | (2 * kw == W), 1.0, 2.0)[None, :]
    G4 = np.concatenate([ck * np.cos(aw).T, -ck * np.sin(aw).T], axis=1) / math.sqrt(W)   # [W, (re|im, kw)]
    kh, h = np.arange(r0, r1), np.arange(H)
    ah = 2.0 * np.pi * np.outer(kh, h) / H                        # [R, H]
    c, s_ = np.cos(ah) / math.sqrt(H), np.sin(ah) / math.sqrt(H)
    E2 = np.empty((2, R, 2 * H))                                  # [out plane][kh][(h, in plane)]
    E2[0, :, 0::2], E2[0, :, 1::2] = c, s_                        # Re: cos*Tr + sin*Ti
    E2[1, :, 0::2], E2[1, :, 1::2] = -s_, c                       # Im: cos*Ti - sin*Tr
    E3 = np.empty((2 * H, 2 * R))                                 # [(h, out plane)][(in plane, kh)]
    E3[0::2, :R], E3[0::2, R:] = c.T, -s_.T                       # Re: cos*Yr - sin*Yi
    E3[1::2, :R], E3[1::2, R:] = s_.T, c.T                        # Im: sin*Yr + cos*Yi
    t = tuple(torch.from_numpy(np.ascontiguousarray(a)).float().to(device) for a in (F1, E2, E3, G4))
    _TABLES[key] = (r0, r1, c1) + t
    return _TABLES[key]


class _BlockComplexLinear(torch.autograd.Function):
    """O[ro][t][blk, o] = act( sum_ri sum_i X[ri][t][blk, i] * wq[ri][ro][blk][i][o] + b[ro][blk][o] ):
    the complex block-diagonal layer of the AFNO mixer on planar (re | im) spectra X [2, T, C]."""

    @staticmethod
    def forward(ctx, X, w, b, act, lam):
        lib = L.load()
        _, T, C = X.shape
        _, nb, bsi, bso = w.shape
        assert nb * bsi == C
        Co = nb * bso
        n = nb * bsi * bso
        X = X.contiguous()
        wq = torch.empty(2, 2, nb, bsi, bso, device=X.device)
        L.check(lib.dlwp_afno_wq_expand(L.ptr(w.contiguous()), L.ptr(wq), nb, bsi, bso, L.stream()))
        O = torch.empty(2, T, Co, device=X.device)
        P = torch.empty(2, T, Co, device=X.device) if act else None
        common = dict(M=T, N=bso, K=bsi, lda=C, ldb=bso, ldc=Co, tA=0, tB=0, nb1=nb, nb2=2, sA=(bsi, 0), sB=(bsi * bso, n),
                      sC=(bso, T * Co))
        _gemm_batched(X, wq, O, **common, oA=0, oB=0)
        _gemm_batched(X, wq, O, **common, oA=T * C, oB=2 * n, bias=b.contiguous(), sBi=(bso, nb * bso), act=act, act_param=lam,
                      preact=P, residual=O, sR=(bso, T * Co), res_pre=1)
        ctx.save_for_backward(X, wq, P)
        ctx.cfg = (T, C, Co, nb, bsi, bso, n, act, lam)
        ctx.slots = (_grad_slot(w), _grad_slot(b))
        ctx.shapes = (w.shape, b.shape)
        return O

    @staticmethod
    def backward(ctx, gO):
        lib = L.load()
        X, wq, P = ctx.saved_tensors
        T, C, Co, nb, bsi, bso, n, act, lam = ctx.cfg
        gO = gO.contiguous()
        if act:
            gP = torch.empty_like(gO)
            L.check(lib.dlwp_act_bwd(L.ptr(P), L.ptr(gO), L.ptr(gP), gO.numel(), act, lam, L.stream()))
        else:
            gP = gO
        # gX[ri] = sum_ro gP[ro] . wq[ri][ro]^T
        gX = torch.empty_like(X)
        for ro in range(2):
            _gemm_batched(gP, wq, gX, T, bsi, bso, Co, bso, C, 0, 1, nb, 2, (bso, 0), (bsi * bso, 2 * n), (bsi, T * C),
                          accumulate=ro, oA=ro * T * Co, oB=ro * n)
        # gwq[ri][ro][blk] = X[ri][:, blk]^T . gP[ro][:, blk]   (K = tokens: split along K inside the kernel)
        gwq = torch.zeros_like(wq)        # one fill; the split-K partial sums are accumulated with atomics
        for ri in range(2):
            _gemm_batched(X, gP, gwq, bsi, bso, T, C, Co, bso, 1, 0, nb, 2, (bsi, 0), (bso, T * Co), (bsi * bso, n),
                          accumulate=1, oA=ri * T * C, oC=ri * 2 * n)
        wslot, bslot = ctx.slots
        gw = wslot if wslot is not None else torch.zeros(ctx.shapes[0], device=X.device)
        L.check(lib.dlwp_afno_wq_fold(L.ptr(gwq), L.ptr(gw), nb, bsi, bso, L.stream()))
        gb = bslot if bslot is not None else torch.zeros(ctx.shapes[1], device=X.device)
        for ro in range(2):
            L.check(lib.dlwp_colsum(L.ptr(gP) + 4 * ro * T * Co, L.ptr(gb) + 4 * ro * Co, T, Co, L.stream()))
        return gX, (None if wslot is not None else gw), (None if bslot is not None else gb), None, None


def _bp_layer_forward(X, w, b, act, lam):
    """One complex block-diagonal layer on block-planar spectra X [T, 2 C] (tokens [T][blk][re | im][bs_in],
    fft.rfft2_planar(block=bs_in)): with the weights as one real [2 bs_in, 2 bs_out] matrix per channel block
    (dlwp_afno_wq_expand_bp) the layer is ONE batched GEMM with bias, activation and stored pre-activation in its epilogue (the
    planar form above takes two accumulating passes per product, one per input plane).  Returns (O, wq, P)."""
    lib = L.load()
    T = X.shape[0]
    _, nb, bsi, bso = w.shape
    C, Co = nb * bsi, nb * bso
    wq = torch.empty(nb, 2 * bsi, 2 * bso, device=X.device)
    bq = torch.empty(2 * Co, device=X.device)
    L.check(lib.dlwp_afno_wq_expand_bp(L.ptr(w.contiguous()), L.ptr(b.contiguous()), L.ptr(wq), L.ptr(bq), nb, bsi, bso, L.stream()))
    O = torch.empty(T, 2 * Co, device=X.device, dtype=X.dtype)          # (bf16 spectra under bf16 storage: same type throughout)
    P = torch.empty(T, 2 * Co, device=X.device, dtype=X.dtype) if act else None
    _gemm_batched(X, wq, O, T, 2 * bso, 2 * bsi, 2 * C, 2 * bso, 2 * Co, 0, 0, nb, 1, (2 * bsi, 0), (4 * bsi * bso, 0), (2 * bso, 0),
                  bias=bq, sBi=(2 * bso, 0), act=act, act_param=lam, preact=P)
    return O, wq, P


def _bp_layer_backward(X, wq, P, gO, act, lam, wshape, bshape, wslot, bslot, prev=None):
    """Gradients of _bp_layer_forward: one batched GEMM each for gX and the block weights, one column sum for the bias; the
    parameter gradients are ACCUMULATED into the flat-gradient slots when the parameters have them.  act = 0 with P None: gO is
    already the gradient of the pre-activation.  prev = (act_prev, lam_prev, P_prev): X was act_prev(P_prev) -- the input-gradient
    GEMM then multiplies by that activation's derivative in its epilogue (act 5 / 6 of dlwp_gemm_batched) and the returned gX is
    the gradient of P_prev.  Returns (gX, gw, gb) with gw / gb None when they went into a slot."""
    lib = L.load()
    T = X.shape[0]
    _, nb, bsi, bso = wshape
    C, Co = nb * bsi, nb * bso
    if act:
        if gO.dtype != torch.float32:
            raise L.DlwpError("afno block layer backward: the separate activation pass takes fp32 spectra (bf16 spectra use the masked store)")
        gP = torch.empty_like(gO)
        L.check(lib.dlwp_act_bwd(L.ptr(P), L.ptr(gO), L.ptr(gP), gO.numel(), act, lam, L.stream()))
    else:
        gP = gO
    gX = torch.empty_like(X)                 # gX[:, blk] = gP[:, blk] . wq[blk]^T
    extra = {}
    if prev is not None:
        act_prev, lam_prev, P_prev = prev
        extra = dict(act={2: 5, 3: 6}[act_prev], act_param=lam_prev, residual=P_prev, sR=(2 * bsi, 0))
    _gemm_batched(gP, wq, gX, T, 2 * bsi, 2 * bso, 2 * Co, 2 * bso, 2 * C, 0, 1, nb, 1, (2 * bso, 0), (4 * bsi * bso, 0), (2 * bsi, 0),
                  **extra)
    nq = nb * 4 * bsi * bso
    gq = torch.zeros(nq + 2 * Co, device=X.device)      # [gwq | gbq]: one fill; split-K partial sums / column sums accumulate
    # gwq[blk] = X[:, blk]^T . gP[:, blk]   (K = tokens: split along K inside the kernel)
    _gemm_batched(X, gP, gq, 2 * bsi, 2 * bso, T, 2 * C, 2 * Co, 2 * bso, 1, 0, nb, 1, (2 * bsi, 0), (2 * bso, 0), (4 * bsi * bso, 0),
                  accumulate=1)
    if gP.dtype == torch.bfloat16:
        L.check(lib.dlwp_colsum_bf16(L.ptr(gP), L.ptr(gq) + 4 * nq, T, 2 * Co, L.stream()))
    else:
        L.check(lib.dlwp_colsum(L.ptr(gP), L.ptr(gq) + 4 * nq, T, 2 * Co, L.stream()))
    gw = wslot if wslot is not None else torch.zeros(wshape, device=X.device)
    gb = bslot if bslot is not None else torch.zeros(bshape, device=X.device)
    L.check(lib.dlwp_afno_wq_fold_bp(L.ptr(gq), L.ptr(gq) + 4 * nq, L.ptr(gw), L.ptr(gb), nb, bsi, bso, L.stream()))
    return gX, (None if wslot is not None else gw), (None if bslot is not None else gb)


# env: A/B runs of the soft-shrink derivative inside the adjoint transform's store (round 5) against a dlwp_act_bwd pass
_MASKED_R2C = __import__("os").environ.get("DLWP_AFNO_MASKED_R2C", "1") != "0"
# env: opt-in bf16 spectrum window under bf16 storage (round 5: +2 % on C5).  OFF by default since round 6: the reference computes the
# spectral mixer in fp32 (fourcastnet.py:80-81,120-124: x.float() before rfft2, softshrink at 0.01 on fp32 values, cast back after
# irfft2) and tools/bf16_training_quality.py could not show that rounding the window to bf16 leaves the trained error unchanged
_SPECTRA_BF16 = __import__("os").environ.get("DLWP_AFNO_SPECTRA_BF16", "0") != "0"


class _AfnoFftFilterFn(torch.autograd.Function):
    """AFNO2D on the FFT path as ONE autograd node: rfft2 (kept window, block-planar) -> complex block MLP (ReLU, soft-shrink) ->
    irfft2 + x, the skip added in the inverse transform's store; backward mirrors it and adds the gradient that arrived along the
    skip in the store of the last transform -- no separate residual add, no autograd accumulation at the fork of x
    (reference: AFNO2D.forward, src/dlwpbench/models/fourcastnet/fourcastnet.py:77-126)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, win, lam, residual=None):
        """residual: the block's outer skip (fourcastnet.py:156-165), added in the same store as the filter's own `+ x`"""
        from . import fft
        B, H, W, C = x.shape
        r0, r1, c1 = win
        bs = w1.shape[2]
        x = x.contiguous().float()
        # DLWP_AFNO_SPECTRA_BF16=1 under bf16 storage: the spectrum window and every operand of the block MLP are bf16 arrays
        # (transforms, accumulation and epilogues stay fp32).  Default: the window stays fp32 as in the reference (no autocast there)
        lowp = _SPECTRA_BF16 and _act_dtype() == torch.bfloat16 and _MASKED_R2C
        X = fft._run_r2c_planar(x, win, bs, fft.NORMS["ortho"], 0, out_bf16=lowp).view(B * (r1 - r0) * c1, 2 * C)
        O1, wq1, P1 = _bp_layer_forward(X, w1, b1, 2, 0.0)
        O2, wq2, P2 = _bp_layer_forward(O1, w2, b2, 3, lam)
        res2 = residual.reshape(x.shape).contiguous().float() if residual is not None else None
        y = fft._run_c2r_planar(O2.view(B, r1 - r0, c1, C // bs, 2, bs), H, W, win, bs, fft.NORMS["ortho"], 0, residual=x, residual2=res2)
        ctx.has_res = residual is not None
        ctx.save_for_backward(X, wq1, P1, O1, wq2, P2)
        ctx.cfg = (B, H, W, C, win, bs, lam, w1.shape, b1.shape, w2.shape, b2.shape)
        ctx.slots = tuple(_grad_slot(p) for p in (w1, b1, w2, b2))
        return y

    @staticmethod
    def backward(ctx, gy):
        from . import fft
        X, wq1, P1, O1, wq2, P2 = ctx.saved_tensors
        B, H, W, C, win, bs, lam, w1s, b1s, w2s, b2s = ctx.cfg
        r0, r1, c1 = win
        gy = gy.contiguous().float()
        if _MASKED_R2C:
            # the adjoint transform's store applies softshrink'(P2): gP2 leaves it directly (no activation-backward pass)
            gP2 = fft._run_r2c_planar(gy, win, bs, fft.NORMS["ortho"], 1, mask=P2, lam=lam,
                                      out_bf16=P2.dtype == torch.bfloat16).view(X.shape[0], -1)
            gP1, gw2, gb2 = _bp_layer_backward(O1, wq2, None, gP2, 0, lam, w2s, b2s, ctx.slots[2], ctx.slots[3], prev=(2, 0.0, P1))
        else:
            gO2 = fft._run_r2c_planar(gy, win, bs, fft.NORMS["ortho"], 1).view(X.shape[0], -1)
            # layer 2's input-gradient GEMM multiplies by ReLU'(P1) in its epilogue: gP1 leaves it directly
            gP1, gw2, gb2 = _bp_layer_backward(O1, wq2, P2, gO2, 3, lam, w2s, b2s, ctx.slots[2], ctx.slots[3], prev=(2, 0.0, P1))
        gX, gw1, gb1 = _bp_layer_backward(X, wq1, None, gP1, 0, 0.0, w1s, b1s, ctx.slots[0], ctx.slots[1])
        gx = fft._run_c2r_planar(gX.view(B, r1 - r0, c1, C // bs, 2, bs), H, W, win, bs, fft.NORMS["ortho"], 1, residual=gy)
        return gx, gw1, gb1, gw2, gb2, None, None, (gy if ctx.has_res else None)


def afno2d_tiled(x, w1, b1, w2, b2, num_blocks, sparsity_threshold=0.01, hard_thresholding_fraction=1.0):
    """x [B, H, W, C] -> AFNO2D(x) including the residual `+ x`."""
    B, H, W, C = x.shape
    x = x.contiguous().float()
    r0, r1, c1, F1, E2, E3, G4 = _tables(H, W, hard_thresholding_fraction, x.device)
    R = r1 - r0
    if B * H > 65535:
        raise L.DlwpError("afno2d_tiled: B*H > 65535 rows per call; split the batch")
    g1 = dict(M=2 * c1, N=C, K=W, lda=W, tA=0, ldx=C, ldy=C, nb1=B * H, nb2=1, sA=(0, 0), sX=(W * C, 0), sY=(2 * c1 * C, 0),
              out_shape=(B, H, 2, c1, C))
    t1 = _TableGemm.apply(x, F1, g1, None)
    g2 = dict(M=R, N=c1 * C, K=2 * H, lda=2 * H, tA=0, ldx=c1 * C, ldy=c1 * C, nb1=B, nb2=2, sA=(0, R * 2 * H),
              sX=(2 * H * c1 * C, 0), sY=(R * c1 * C, B * R * c1 * C), out_shape=(2, B, R, c1, C))
    X = _TableGemm.apply(t1, E2, g2, None)
    T = B * R * c1
    o1 = _BlockComplexLinear.apply(X.view(2, T, C), w1, b1, 2, 0.0)
    o2 = _BlockComplexLinear.apply(o1, w2, b2, 3, float(sparsity_threshold))
    g3 = dict(M=2 * H, N=c1 * C, K=R, lda=2 * R, tA=0, ldx=c1 * C, ldy=c1 * C, nb1=B, nb2=1, sA=(0, 0), sX=(R * c1 * C, 0),
              sY=(2 * H * c1 * C, 0), out_shape=(B, H, 2, c1, C), passes=((0, 0), (R, B * R * c1 * C)))
    t2 = _TableGemm.apply(o2.view(2, B, R, c1, C), E3, g3, None)
    g4 = dict(M=W, N=C, K=2 * c1, lda=2 * c1, tA=0, ldx=C, ldy=C, nb1=B * H, nb2=1, sA=(0, 0), sX=(2 * c1 * C, 0),
              sY=(W * C, 0), out_shape=(B, H, W, C))
    return _TableGemm.apply(t2, G4, g4, x)


def afno2d_fft(x, w1, b1, w2, b2, num_blocks, sparsity_threshold=0.01, hard_thresholding_fraction=1.0, residual=None):
    """AFNO2D on the LDS-staged rFFT2 / irFFT2 kernels (fft.py, csrc/fft2d.hip) instead of dense DFT GEMMs: linear-ish in the
    grid size, so this is the path of patch-1 grids (dlwpbench fourcastnet.yaml: patch_size [1, 1]; 128 x 256, 721 x 1440).
    x [B, H, W, C] -> AFNO2D(x) including the residual `+ x`; the block-diagonal complex MLP on the kept modes is the same
    batched-GEMM node as in the tiled path (planar re | im spectra); the transforms themselves read / write the kept window in
    that layout (fft.rfft2_planar / irfft2_planar)."""
    from . import fft
    from .token_ops import add_tokens
    B, H, W, C = x.shape
    x = x.contiguous().float()
    r0, r1, c1 = kept_window(H, W, hard_thresholding_fraction)
    R = r1 - r0
    # the transforms read / write the kept window in the planar layout of the block GEMMs: no window copies, no zero fill, and
    # the H pass of the forward transform skips the columns outside the window
    win = (r0, r1, c1)
    nb, bs = w1.shape[1], w1.shape[2]
    if bs >= 2 and w2.shape[3] == bs:
        # block-planar spectra: one batched GEMM per layer and gradient, the whole filter one autograd node
        return _AfnoFftFilterFn.apply(x, w1, b1, w2, b2, win, float(sparsity_threshold), residual)
    planar = fft.rfft2_planar(x, "ortho", win).view(2, B * R * c1, C)
    o1 = _BlockComplexLinear.apply(planar, w1, b1, 2, 0.0)
    o2 = _BlockComplexLinear.apply(o1, w2, b2, 3, float(sparsity_threshold))
    y = add_tokens(fft.irfft2_planar(o2.view(2, B, R, c1, C), H, W, "ortho", win), x)
    return y if residual is None else add_tokens(y, residual)
