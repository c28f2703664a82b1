"""dlwpbench AFNONet / FourCastNet on libdlwpmi: same constructor keys, forward signature and state_dict keys as
src/dlwpbench/models/fourcastnet/fourcastnet.py:215-361 (`FourCastNet = AFNONet`, models/__init__.py:4-12).

The network body is the nsbench one (AFNO2D mixer kernel, LayerNorm / MLP / patch-embedding GEMMs, see
../nsbench/fourcastnet.py); what differs is the input assembly (constants + prescribed + prognostic channels),
the optional position embedding and the dlwpbench rollout (rollout.py).
"""
from functools import partial

import torch
import torch.nn as nn

from ..nsbench.fourcastnet import Block, PatchEmbed
from ..token_ops import LayerNorm, Linear, add_pos_embed
from .rollout import rollout


class AFNONet(nn.Module):
    def __init__(self, img_height=720, img_width=1440, patch_size=(16, 16), constant_channels: int = 4,
                 prescribed_channels: int = 0, prognostic_channels: int = 1, filter="AFNO2D", embed_dim=768, depth=12,
                 mlp_ratio=4., drop_rate=0., drop_path_rate=0., num_blocks=16, sparsity_threshold=0.01,
                 hard_thresholding_fraction=1.0, context_size: int = 1, use_pos_embed: bool = True, **kwargs):
        super().__init__()
        if filter != "AFNO2D":
            raise NotImplementedError("only the AFNO2D filter exists in the reference file (fourcastnet.py:77)")
        if drop_rate or drop_path_rate:
            raise NotImplementedError("dropout / stochastic depth are not on the MI355X hot path (configs use 0.0)")
        self.img_size, self.patch_size = (img_height, img_width), tuple(patch_size)
        self.in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.out_chans = prognostic_channels
        self.num_features = self.embed_dim = embed_dim
        self.num_blocks, self.context_size, self.use_pos_embed = num_blocks, context_size, use_pos_embed
        norm_layer = partial(LayerNorm, eps=1e-6)
        self.patch_embed = PatchEmbed(self.img_size, self.patch_size, self.in_chans, embed_dim)
        if use_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        self.h, self.w = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        self.blocks = nn.ModuleList([
            Block(dim=embed_dim, mlp_ratio=mlp_ratio, drop=drop_rate, drop_path=0.0, norm_layer=norm_layer,
                  num_blocks=num_blocks, sparsity_threshold=sparsity_threshold,
                  hard_thresholding_fraction=hard_thresholding_fraction) for _ in range(depth)])
        self.norm = norm_layer(embed_dim)   # constructed but unused, as in the reference (:263, :287-297)
        self.head = Linear(embed_dim, self.out_chans * self.patch_size[0] * self.patch_size[1], bias=False)
        if use_pos_embed:
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
        x = self.patch_embed(x)
        if self.use_pos_embed:
            x = add_pos_embed(x, self.pos_embed)
        x = x.reshape(B, self.h, self.w, self.embed_dim)
        for blk in self.blocks:
            x = blk(x)
        return x

    def forward_one_step(self, x):
        """[B, in_chans, H, W] -> increment [B, prognostic_channels, H, W] (head + un-patchify, :344-357)."""
        B = x.shape[0]
        ph, pw = self.patch_size
        t = self.head(self.forward_features(x))
        t = t.reshape(B, self.h, self.w, ph, pw, self.out_chans).permute(0, 5, 1, 3, 2, 4)
        return t.reshape(B, self.out_chans, self.h * ph, self.w * pw)

    def forward(self, constants: torch.Tensor = None, prescribed: torch.Tensor = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        # the head's patch tokens go to the rollout as they are: un-patching rides the window-advance kernel
        return rollout(lambda x: self.head(self.forward_features(x)), self.context_size, constants, prescribed, prognostic,
                       patch=self.patch_size)


FourCastNet = AFNONet


class SFNONet(nn.Module):
    """FourCastNetv2 = SFNONet (src/dlwpbench/models/fourcastnet/fourcastnet.py:364-527, models/__init__.py:7): patch
    embedding -> (+ pos_embed) -> SFNO (embed_dim -> embed_dim, see sfno.py; third-party network, PARITY UNPINNED) ->
    head -> un-patchify, in the dlwpbench rollout.  The reference builds the inner SFNO for img_size = (img_height,
    img_width) but feeds it the patch grid (:411-415, :455-457), so only patch_size (1, 1) is shape-consistent there;
    here the SFNO is built for the patch grid, which coincides with the reference whenever the reference runs."""

    def __init__(self, img_height=720, img_width=1440, patch_size=(16, 16), constant_channels: int = 4,
                 prescribed_channels: int = 0, prognostic_channels: int = 1, spectral_transform="sht", grid="legendre-gauss",
                 num_layers=4, scale_factor=3, embed_dim=768, operator_type="driscoll-healy", drop_rate=0., num_blocks=16,
                 hard_thresholding_fraction=1.0, factorization: str = None, rank: float = 1.0, big_skip: bool = False,
                 use_pos_embed: bool = True, use_mlp: bool = False, normalization_layer: str = None, context_size: int = 1,
                 **kwargs):
        super().__init__()
        from .sfno import SFNO
        if drop_rate:
            raise NotImplementedError("dropout is not on the MI355X hot path (configs use 0.0)")
        self.img_size, self.patch_size = (img_height, img_width), tuple(patch_size)
        self.in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.out_chans = prognostic_channels
        self.num_features = self.embed_dim = embed_dim
        self.context_size, self.use_pos_embed = context_size, use_pos_embed
        self.patch_embed = PatchEmbed(self.img_size, self.patch_size, self.in_chans, embed_dim)
        if use_pos_embed:
            self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_embed.num_patches, embed_dim))
        self.h, self.w = self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1]
        self.sfno = SFNO(in_chans=embed_dim, out_chans=embed_dim, spectral_transform=spectral_transform,
                         img_size=(self.h, self.w), grid=grid, num_layers=num_layers, scale_factor=scale_factor,
                         embed_dim=embed_dim, operator_type=operator_type,
                         hard_thresholding_fraction=hard_thresholding_fraction, factorization=factorization, rank=rank,
                         big_skip=big_skip, pos_embed=use_pos_embed, use_mlp=use_mlp, normalization_layer=normalization_layer)
        self.norm = LayerNorm(embed_dim, eps=1e-6)   # constructed but unused, as in the reference (:430, :449-458)
        self.head = Linear(embed_dim, self.out_chans * self.patch_size[0] * self.patch_size[1], bias=False)
        if use_pos_embed:
            nn.init.trunc_normal_(self.pos_embed, std=.02)
        nn.init.trunc_normal_(self.head.weight, std=.02)

    def forward_features(self, x):
        B = x.shape[0]
        x = self.patch_embed(x)
        if self.use_pos_embed:
            x = add_pos_embed(x, self.pos_embed)
        x = x.reshape(B, self.h, self.w, self.embed_dim).permute(0, 3, 1, 2)
        return self.sfno(x).permute(0, 2, 3, 1)

    def forward_one_step(self, x):
        B = x.shape[0]
        ph, pw = self.patch_size
        t = self.head(self.forward_features(x))
        t = t.reshape(B, self.h, self.w, ph, pw, self.out_chans).permute(0, 5, 1, 3, 2, 4)
        return t.reshape(B, self.out_chans, self.h * ph, self.w * pw)

    def forward(self, constants: torch.Tensor = None, prescribed: torch.Tensor = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        return rollout(self.forward_one_step, self.context_size, constants, prescribed, prognostic)


FourCastNetv2 = SFNONet
