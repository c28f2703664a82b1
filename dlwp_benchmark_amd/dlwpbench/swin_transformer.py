"""dlwpbench SwinTransformer on libdlwpmi: constructor keys, forward signature and state_dict keys of
src/dlwpbench/models/swintransformer/swin_transformer.py:494-737.

Layers are the nsbench ones (../nsbench/swin_transformer.py: fused window attention, LayerNorm / MLP / merging GEMMs)
with (h, w) window pairs: every stage attends over its whole feature map (window = (H/p, W/p) halved per stage,
reference :542-571) with a half-map cyclic shift in the odd blocks.  Grids that would need window padding are refused:
the reference pads the wrong axes there (:218-222) and its BasicLayer raises.  Patch embedding pads
longitude circularly and latitude with zeros (:446-451).  The rollout is the dlwpbench loop in its working form
(rollout.py).
"""
import torch
import torch.nn as nn

from ..nsbench.swin_transformer import _NORMS, BasicLayer, PatchEmbed, PatchMerging, absolute_position_tokens
from ..token_ops import DropPathPool, PatchConv2d, UpConvT2d
from .rollout import rollout


class SwinTransformer(nn.Module):
    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 context_size: int = 1, img_height=224, img_width=196, patch_size=4, embed_dim=96, depths=[2, 2, 6, 2],
                 num_heads=[3, 6, 12, 24], mlp_ratio=4., qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0.,
                 drop_path_rate=0.2, norm_layer="nn.LayerNorm", ape=False, patch_norm=True, frozen_stages=-1,
                 use_checkpoint=False, mesh="equirectangular", window_size=None, **kwargs):
        """window_size (extra kwarg, not in the reference): None = the reference behaviour (whole-map windows); an int or
        (h, w) pair gives classic Swin windows with the CORRECT per-axis padding (constant latitude, circular longitude) --
        the reference's own block cannot run that case."""
        super().__init__()
        if mesh != "equirectangular":
            raise NotImplementedError("only the equirectangular mesh is on the MI355X hot path (healpix needs dgl)")
        if frozen_stages >= 0:
            raise NotImplementedError("frozen_stages >= 0 (a fine-tuning option: stop gradients of the first stages) is not "
                                      "built; the shipped configs use -1")
        if drop_rate or attn_drop_rate:
            raise NotImplementedError("dropout is not on the MI355X hot path (the shipped config uses drop_rate 0 and "
                                      "attn_drop_rate 0)")
        norm = _NORMS[norm_layer] if isinstance(norm_layer, str) else norm_layer
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]    # stochastic depth decay rule (:552)
        self.context_size, self.num_layers, self.embed_dim = context_size, len(depths), embed_dim
        self.img_height, self.img_width, self.mesh = img_height, img_width, mesh
        in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        pad_modes = ("constant", "circular")     # (latitude, longitude)
        self.patch_embed = PatchEmbed(patch_size, in_chans, embed_dim, norm if patch_norm else None, pad_modes)
        res = (img_height // patch_size, img_width // patch_size)
        self.ape = ape
        if ape:     # learned [1, E, Wh0, Ww0] embedding added to the embedded patches (reference :540-547)
            self.absolute_pos_embed = nn.Parameter(torch.zeros(1, embed_dim, res[0], res[1]))
            nn.init.trunc_normal_(self.absolute_pos_embed, std=.02)
        self.layers = nn.ModuleList()
        for i in range(self.num_layers):
            if window_size is None and i < self.num_layers - 1 and (res[0] % 2 or res[1] % 2):
                raise NotImplementedError(f"stage {i} feature map {res} is odd: the reference's window padding is broken "
                                          "there (swin_transformer.py:218-222, SURVEY App. B-6)")
            self.layers.append(BasicLayer(int(embed_dim * 2 ** i), depths[i], num_heads[i],
                                          res if window_size is None else window_size, mlp_ratio, qkv_bias,
                                          qk_scale, drop_path=dpr[sum(depths[:i]):sum(depths[:i + 1])], norm_layer=norm,
                                          downsample=PatchMerging if i < self.num_layers - 1 else None,
                                          padding_mode=pad_modes))
            res = (res[0] // 2, res[1] // 2)
        self.num_features = [int(embed_dim * 2 ** i) for i in range(self.num_layers)]
        for i, nf in enumerate(self.num_features):
            self.add_module(f"norm{i}", norm(nf))
        self.decoder = nn.ModuleList()
        for idx, i in enumerate(reversed(range(self.num_layers))):
            ch = int(embed_dim * 2 ** i)
            k = patch_size if i == 0 else 2
            self.decoder.append(nn.Sequential(
                UpConvT2d(ch if idx == 0 else 2 * ch, ch if i == 0 else ch // 2, kernel_size=k, stride=k), nn.GELU()))
        self.final = PatchConv2d(embed_dim, prognostic_channels, kernel_size=1)

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

    def forward(self, constants: torch.Tensor = None, prescribed: torch.Tensor = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        return rollout(self.one_step, self.context_size, constants, prescribed, prognostic)
