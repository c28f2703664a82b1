"""Drop-in counterpart of the reference's FourCastNet (AFNO) rollout model.

Reference (file:line under /root/reference/src/nsbench/models/fourcastnet/fourcastnet.py):
  AFNO2D :59-126, Mlp :40-56, Block :129-165, PatchEmbed :303-316, AFNONet :186-300.
Same constructor kwargs, `forward(x[B,T,D,H,W], teacher_forcing_steps)`, parameter names and shapes
(state_dict keys patch_embed.proj, pos_embed, blocks.{i}.{norm1,filter.{w1,b1,w2,b2},norm2,mlp.{fc1,fc2}},
norm, head) so reference checkpoints load.

The spectral token mixer (AFNO2D: rfft2 -> block-diagonal complex MLP -> softshrink -> irfft2 ->
residual) runs as one hand-written HIP kernel per direction (libdlwpmi dlwp_afno2d_fwd/bwd).
LayerNorm, the token MLP (GELU and residual fused in the GEMM epilogues) and the head run on libdlwpmi's
MFMA GEMM / LayerNorm kernels (token_ops.py).  The patch embedding is an unfold + the same GEMM (PatchConv2d).
"""
from functools import partial

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import lib as L
from ..rollout_ops import ns_rollout
from ..token_ops import LayerNorm, Linear, Mlp, PatchConv2d, WgradBatch, _grad_slot, add_pos_embed, add_tokens, norm_fork

_AFNO_WGRAD_BATCH = __import__("os").environ.get("DLWP_AFNO_WGRAD_BATCH", "0") == "1"
_AFNO_RES2 = __import__("os").environ.get("DLWP_AFNO_RES2", "1") != "0"


FFT_MIN_TOKENS = 262144     # token grids from this size on take the rFFT2 path in "auto" mode (see AFNO2D.forward)


class _AFNO2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, nb, lam, frac, residual=None):
        lib = L.load()
        B, H, W, C = x.shape
        x = x.contiguous().float()
        res = residual.reshape(x.shape).contiguous().float() if residual is not None else None
        n = lib.dlwp_afno2d_save_elems(B, H, W, C, nb, frac)
        if n < 0:
            L.check(-3)
        xsave = torch.empty(n, device=x.device)
        y = torch.empty_like(x)
        L.check(lib.dlwp_afno2d_fwd_res(L.ptr(x), L.ptr(res), L.ptr(w1.contiguous()), L.ptr(b1.contiguous()),
                                        L.ptr(w2.contiguous()), L.ptr(b2.contiguous()), L.ptr(y), L.ptr(xsave), B, H, W, C, nb,
                                        lam, frac, L.stream()))
        ctx.save_for_backward(xsave, w1, b1, w2, b2)
        ctx.cfg = (B, H, W, C, nb, lam, frac)
        ctx.has_res = residual is not None
        ctx.slots = [_grad_slot(t) for t in (w1, b1, w2, b2)]
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = L.load()
        xsave, w1, b1, w2, b2 = ctx.saved_tensors
        B, H, W, C, nb, lam, frac = ctx.cfg
        gy = gy.contiguous().float()
        gx = torch.empty_like(gy)
        fused = all(sl is not None for sl in ctx.slots)    # kernel accumulates: write straight into .grad
        gw1, gb1, gw2, gb2 = ctx.slots if fused else [torch.zeros_like(t, memory_format=torch.contiguous_format)
                                                      for t in (w1, b1, w2, b2)]
        L.check(lib.dlwp_afno2d_bwd(L.ptr(gy), L.ptr(xsave), L.ptr(w1.contiguous()), L.ptr(b1.contiguous()),
                                    L.ptr(w2.contiguous()), L.ptr(b2.contiguous()), L.ptr(gx), L.ptr(gw1), L.ptr(gb1),
                                    L.ptr(gw2), L.ptr(gb2), B, H, W, C, nb, lam, frac, L.stream()))
        gres = gy if ctx.has_res else None         # the skip's gradient is the upstream gradient itself
        if fused:
            return gx, None, None, None, None, None, None, None, gres
        return gx, gw1, gb1, gw2, gb2, None, None, None, gres


class AFNO2D(nn.Module):
    def __init__(self, hidden_size, num_blocks=8, sparsity_threshold=0.01, hard_thresholding_fraction=1,
                 hidden_size_factor=1):
        super().__init__()
        assert hidden_size % num_blocks == 0, f"hidden_size {hidden_size} should be divisble by num_blocks {num_blocks}"
        if hidden_size_factor != 1:
            raise NotImplementedError("hidden_size_factor != 1 is not on the MI355X hot path")
        # "fused" (LDS-resident kernel), "tiled" (dense DFT as batched GEMMs, any grid), "fft" (LDS-staged rFFT2 kernels,
        # any grid, even channel count) or "auto"
        self.path = "auto"
        self.hidden_size, self.num_blocks = hidden_size, num_blocks
        self.block_size = hidden_size // num_blocks
        self.sparsity_threshold = sparsity_threshold
        self.hard_thresholding_fraction = hard_thresholding_fraction
        self.scale = 0.02
        bs = self.block_size
        self.w1 = nn.Parameter(self.scale * torch.randn(2, num_blocks, bs, bs))
        self.b1 = nn.Parameter(self.scale * torch.randn(2, num_blocks, bs))
        self.w2 = nn.Parameter(self.scale * torch.randn(2, num_blocks, bs, bs))
        self.b2 = nn.Parameter(self.scale * torch.randn(2, num_blocks, bs))

    def forward(self, x, residual=None):
        """AFNO2D(x) (+ residual: the block's outer skip, added by the fused kernel's epilogue)."""
        dtype = x.dtype
        B, H, W, C = x.shape
        path = self.path
        if path == "auto":      # the fused kernel keeps a block's half spectrum in LDS: small grids, block size <= 16
            fits = L.load().dlwp_afno2d_save_elems(B, H, W, C, self.num_blocks, float(self.hard_thresholding_fraction)) >= 0
            # beyond the LDS-resident kernel (profiles/r02_fft_bench.txt, forward + backward): the dense-DFT GEMMs ride the
            # matrix cores and stay ahead of the FFT path up to 256 x 512 tokens when the grid is a multiple of 4 (aligned
            # 16-byte operand loads); on other grids (103 x 180: 18.5 ms vs 2.05 ms) and on grids where a dense DFT is out
            # of the question (721 x 1440 at patch 1) the rFFT2 kernels take over
            big = H * W >= FFT_MIN_TOKENS or H % 4 != 0 or W % 4 != 0
            path = "fused" if fits else ("fft" if big and C % 2 == 0 else "tiled")
        if path == "fft":
            from ..afno_tiled import afno2d_fft
            # (the block's outer skip rides the inverse transform's store together with the filter's own `+ x`;
            # DLWP_AFNO_RES2=0: a separate add, for A/B runs)
            if _AFNO_RES2:
                return afno2d_fft(x, self.w1, self.b1, self.w2, self.b2, self.num_blocks, float(self.sparsity_threshold),
                                  float(self.hard_thresholding_fraction), residual=residual).type(dtype)
            y = afno2d_fft(x, self.w1, self.b1, self.w2, self.b2, self.num_blocks, float(self.sparsity_threshold),
                           float(self.hard_thresholding_fraction))
        elif path == "tiled":
            from ..afno_tiled import afno2d_tiled
            y = afno2d_tiled(x, self.w1, self.b1, self.w2, self.b2, self.num_blocks, float(self.sparsity_threshold),
                             float(self.hard_thresholding_fraction))
        else:
            return _AFNO2DFn.apply(x, self.w1, self.b1, self.w2, self.b2, self.num_blocks, float(self.sparsity_threshold),
                                   float(self.hard_thresholding_fraction), residual).type(dtype)
        if residual is not None:
            y = add_tokens(y, residual)
        return y.type(dtype)


class Block(nn.Module):
    wgrad_batch_block = True      # the block's weight-gradient writes land together at its first layer's backward (token_ops.WgradBatch)

    def __init__(self, dim, mlp_ratio=4., drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm,
                 double_skip=True, num_blocks=8, sparsity_threshold=0.01, hard_thresholding_fraction=1.0):
        super().__init__()
        if drop_path > 0.:
            raise NotImplementedError("stochastic depth is not on the MI355X hot path (configs use 0.0)")
        self.norm1 = norm_layer(dim)
        self.filter = AFNO2D(dim, num_blocks, sparsity_threshold, hard_thresholding_fraction)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.double_skip = double_skip

    def forward(self, x):
        # reference :156-165.  Both skips are fused: the first into the filter kernel's epilogue, the second into fc2's; the
        # gradients of the skip branches join the LayerNorm backward kernels (LayerNorm.fork)
        if self.double_skip:
            residual, t = norm_fork(self.norm1, x)
            residual, t = norm_fork(self.norm2, self.filter(t, residual=residual), gemm_input=True)
        else:
            residual, t = norm_fork(self.norm1, x)
            t = self.norm2(self.filter(t))
        # (DLWP_AFNO_WGRAD_BATCH=1: the two MLP weight gradients in one dlwp_wgrad_segments launch, measurement knob)
        return self.mlp(t, residual=residual, wbatch=WgradBatch() if _AFNO_WGRAD_BATCH else None)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=(224, 224), patch_size=(16, 16), in_chans=3, embed_dim=768):
        super().__init__()
        self.num_patches = (img_size[1] // patch_size[1]) * (img_size[0] // patch_size[0])
        self.img_size, self.patch_size = img_size, patch_size
        self.proj = PatchConv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)

    def forward(self, x):
        B, C, H, W = x.shape
        assert H == self.img_size[0] and W == self.img_size[1], \
            f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        return self.proj(x).flatten(2).transpose(1, 2)


class AFNONet(nn.Module):
    def __init__(self, img_height=720, img_width=1440, patch_size=(16, 16), in_chans=2, out_chans=2, embed_dim=768,
                 depth=12, mlp_ratio=4., drop_rate=0., drop_path_rate=0., num_blocks=16, sparsity_threshold=0.01,
                 hard_thresholding_fraction=1.0, context_size: int = 1, **kwargs):
        super().__init__()
        self.img_size = (img_height, img_width)
        self.patch_size = tuple(patch_size)
        self.in_chans = in_chans * context_size
        self.out_chans = out_chans
        self.num_features = self.embed_dim = embed_dim
        self.num_blocks = num_blocks
        self.context_size = context_size
        norm_layer = partial(LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(self.img_size, self.patch_size, self.in_chans, embed_dim)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.h = self.img_size[0] // self.patch_size[0]
        self.w = self.img_size[1] // self.patch_size[1]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, mlp_ratio=mlp_ratio, drop=drop_rate, drop_path=0.0, norm_layer=norm_layer,
                  num_blocks=num_blocks, sparsity_threshold=sparsity_threshold,
                  hard_thresholding_fraction=hard_thresholding_fraction) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)  # constructed but unused, as in the reference (:228, :251-261)
        self.head = Linear(embed_dim, self.out_chans * self.patch_size[0] * self.patch_size[1], bias=False)
        nn.init.trunc_normal_(self.pos_embed, std=.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def forward_features(self, x):
        B = x.shape[0]
        x = self.pos_drop(add_pos_embed(self.patch_embed(x), self.pos_embed))
        x = x.reshape(B, self.h, self.w, self.embed_dim)
        for blk in self.blocks:
            x = blk(x)
        return x

    def forward(self, x, teacher_forcing_steps: int = 50):
        # reference :262-300.  The head's patch tokens [B, h*w, ph*pw*out] are un-patched, added to the newest frame and
        # appended to the sliding window by one kernel (rollout_ops.advance)
        return ns_rollout(lambda x_t: self.head(self.forward_features(x_t)), x, teacher_forcing_steps, self.context_size,
                          patch=self.patch_size)


FourCastNet = AFNONet
