"""Drop-in counterpart of the reference's dlwpbench PanguWeather (2-D surface variant).

Reference (file:line under /root/reference/src/dlwpbench/models/panguweather/): EarthAttention3D
panguweather.py:133-211, EarthSpecificBlock :214-323, BasicLayer :326-363, DownSample :80-130, UpSample :30-77,
PanguWeather :366-527; utils/{earth_position_index,shift_window_mask,pad,crop,patch_embed,patch_recovery}.py.
Constructor kwargs, forward(constants, prescribed, prognostic) and parameter names are the reference's.

The earth-specific window attention runs on libdlwpmi's fused attention kernel (scores never reach HBM; the
per-window-type bias index is additive in query and key, so two N-vectors replace the N x N index tensor; the
shift mask is a label vector per window).  LayerNorm / Linear / MLP / patch (de)embedding use token_ops.  The
published multi-step loop is broken (list.to(), out.cpu()): this module keeps predictions on
device (clean form).  The forward roll's longitude shift by the LATITUDE shift (:291) is reproduced.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..nsbench.swin_transformer import _WindowAttnTokensFn, window_attention_core, window_attention_tokens
from ..window_ops import WindowSpec, partition, reverse
from .rollout import rollout
from ..token_ops import DropPath, DropPathPool, LayerNorm, Linear, Mlp, PatchConv2d, UpConvT2d, WgradBatch, norm_fork

_DEFAULT_SHIFT = (1, 3, 6)   # panguweather.py:243


def _pad3d(res, window):
    out = []
    for n, w in ((res[2], window[2]), (res[1], window[1]), (res[0], window[0])):
        r = n % w
        p = (w - r) if r else 0
        out += [p // 2, p - p // 2]
    return tuple(out)   # lon(l, r), lat(t, b), pl(f, b)


def _partition(x, window):
    B, Pl, Lat, Lon, C = x.shape
    a, b, c = window
    x = x.view(B, Pl // a, a, Lat // b, b, Lon // c, c, C).permute(0, 5, 1, 3, 2, 4, 6, 7)
    return x.reshape(B * (Lon // c) * (Pl // a) * (Lat // b), a * b * c, C)


def _unpartition(wins, window, B, Pl, Lat, Lon):
    a, b, c = window
    x = wins.view(B, Lon // c, Pl // a, Lat // b, a, b, c, -1).permute(0, 2, 4, 3, 5, 1, 6, 7)
    return x.reshape(B, Pl, Lat, Lon, -1)


class EarthAttention3D(nn.Module):
    def __init__(self, dim, input_resolution, window_size, num_heads, qkv_bias=True, qk_scale=None, attn_drop=0.,
                 proj_drop=0.):
        super().__init__()
        self.dim, self.window_size, self.num_heads = dim, tuple(window_size), num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        Wpl, Wlat, Wlon = self.window_size
        self.type_of_windows = (input_resolution[0] // Wpl) * (input_resolution[1] // Wlat)
        self.earth_position_bias_table = nn.Parameter(
            torch.zeros(Wpl ** 2 * Wlat ** 2 * (2 * Wlon - 1), self.type_of_windows, num_heads))
        z, h, w = torch.meshgrid(torch.arange(Wpl), torch.arange(Wlat), torch.arange(Wlon), indexing="ij")
        z, h, w = z.reshape(-1), h.reshape(-1), w.reshape(-1)
        s1 = 2 * Wlon - 1
        s0 = s1 * Wlat * Wlat
        ia = z * s0 + h * s1 + w                                   # query part
        ib = z * Wpl * s0 + h * Wlat * s1 + (Wlon - 1 - w)         # key part (earth_position_index.py:4-45)
        self.register_buffer("earth_position_index", ia[:, None] + ib[None, :])   # checkpoint compatibility only
        self.register_buffer("_ia", ia.to(torch.int32), persistent=False)
        self.register_buffer("_ib", ib.to(torch.int32), persistent=False)
        self.qkv = Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = Linear(dim, dim)
        nn.init.trunc_normal_(self.earth_position_bias_table, std=.02)

    def forward(self, x, labels=None, n_windows=1):
        """x [B*n_lon*types, N, C] (window index fastest); labels int32 [n_lon*types, N] or None."""
        y = window_attention_core(self.qkv(x), self.earth_position_bias_table, self._ia, self._ib, labels, n_windows,
                                  self.num_heads, float(self.scale))
        return self.proj(y)

    def core(self, qkv_windows, labels, n_windows, qrange):
        """attention on windows of an already projected qkv tensor [B*nW, N, 3C] (EarthSpecificBlock's real-token flow)"""
        return window_attention_core(qkv_windows, self.earth_position_bias_table, self._ia, self._ib, labels, n_windows,
                                     self.num_heads, float(self.scale), qrange)


class EarthSpecificBlock(nn.Module):
    wgrad_batch_block = True      # the block's weight-gradient writes land together at its first layer's backward (token_ops.WgradBatch)

    def __init__(self, dim, input_resolution, num_heads, window_size=None, shift_size=None, mlp_ratio=4., qkv_bias=True,
                 qk_scale=None, drop=0., attn_drop=0., drop_path=0., act_layer=nn.GELU, norm_layer=LayerNorm):
        super().__init__()
        self.window_size = (2, 6, 12) if window_size is None else tuple(window_size)
        self.shift_size = _DEFAULT_SHIFT if shift_size is None else tuple(shift_size)
        self.input_resolution = tuple(input_resolution)
        self.norm1 = norm_layer(dim)
        self.padding = _pad3d(self.input_resolution, self.window_size)
        p = self.padding
        self.pad_resolution = (self.input_resolution[0] + p[4] + p[5], self.input_resolution[1] + p[2] + p[3],
                               self.input_resolution[2] + p[0] + p[1])
        self.attn = EarthAttention3D(dim, self.pad_resolution, self.window_size, num_heads, qkv_bias, qk_scale)
        self.drop_path = DropPath(drop_path)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio), act_layer=act_layer, drop=drop)
        self.roll = bool(self.shift_size[0] and self.shift_size[1] and self.shift_size[2])
        self._wspec = WindowSpec(self.input_resolution, self.window_size, front=(p[4], p[2], p[0]), back=(p[5], p[3], p[1]),
                                 order=(2, 0, 1))      # windows longitude-major (utils window_partition)
        labels = None
        if self.roll:
            Pl, Lat, Lon = self.pad_resolution
            win, sh = self.window_size, self.shift_size

            def axis(n, wn, s, extra=0):
                lab = torch.zeros(n + extra, dtype=torch.int32)
                lab[n + extra - wn:n + extra - s] = 1
                lab[n + extra - s:] = 2
                return lab[:n]
            # labels of the canvas widened by shift_lon and cropped back (shift_window_mask.py:54-67): windows that
            # straddle the date line are not masked apart
            img = (axis(Pl, win[0], sh[0])[:, None, None] * 9 + axis(Lat, win[1], sh[1])[None, :, None] * 3
                   + axis(Lon, win[2], sh[2], extra=sh[2])[None, None, :])
            labels = _partition(img[None, ..., None].float(), win).reshape(-1, win[0] * win[1] * win[2]).to(torch.int32)
        self.register_buffer("_labels", labels, persistent=False)
        # Padded tokens: the reference zero-pads AFTER norm1 and runs qkv / attention / proj on every window token, then crops
        # (:283-317).  With one pressure level padded to a window of two, half of every window is padding (C4: 68,894 window
        # tokens for 32,768 real ones).  Exactly equivalent and cheaper: qkv on the real tokens, pad the qkv tensor with the qkv
        # BIAS (Linear(0) = bias), attend with the padded tokens as keys / values only (their own rows are cropped: no output
        # needed, zero gradient), crop, then proj on the real tokens.  _qrange = hull of the window positions that hold a real
        # token in ANY window (the query rows worth computing).
        # The rows worth computing are the window positions crop3d KEEPS.  In shifted blocks the reference rolls the longitude
        # forward by the LATITUDE shift and back by the longitude shift (:291 vs :310, reproduced), so the kept positions are
        # not the ones the real tokens were put at: the hull covers both, and the padded positions that are kept receive a
        # non-zero query gradient (-> the fill's adjoint sums all three thirds of the qkv gradient there).
        Pl, Lat, Lon = self.input_resolution
        real = F.pad(torch.ones(1, 1, Pl, Lat, Lon), p).permute(0, 2, 3, 4, 1)
        nwin = self.window_size[0] * self.window_size[1] * self.window_size[2]
        canv = [real]
        if self.roll:
            sh = self.shift_size
            canv = [torch.roll(real, shifts=(-sh[0], -sh[1], -sh[1]), dims=(1, 2, 3)),     # where the tokens go (:291)
                    torch.roll(real, shifts=(-sh[0], -sh[1], -sh[2]), dims=(1, 2, 3))]     # what the crop keeps (:310)
        anyreal = sum(_partition(c, self.window_size).reshape(-1, nwin).amax(0) for c in canv)
        idx = torch.nonzero(anyreal > 0).reshape(-1)
        self._qrange = (int(idx.min()), int(idx.max()) + 1)
        self._kept_are_real = len(canv) == 1 or bool(torch.equal(canv[0], canv[1]))
        n_pad = self.pad_resolution[0] * self.pad_resolution[1] * self.pad_resolution[2]
        self.real_token_flow = n_pad >= 1.25 * Pl * Lat * Lon

    def forward(self, x):
        Pl, Lat, Lon = self.input_resolution
        B, L_, C = x.shape
        assert L_ == Pl * Lat * Lon, "input feature has wrong size"
        p, win, sh = self.padding, self.window_size, self.shift_size
        if C % 4 == 0:
            # ZeroPad3d + roll + partition and reverse + roll back + crop3d as one gather kernel each.  The forward roll
            # uses the latitude shift for the longitude axis, the backward roll the longitude shift (reference :291 vs :310)
            spec = self._wspec
            fwd_shift = (sh[0], sh[1], sh[1]) if self.roll else (0, 0, 0)
            rev_shift = sh if self.roll else (0, 0, 0)
            # skip connections leave the LayerNorm nodes (norm_fork): their gradients join the LayerNorm backward kernels
            wb = WgradBatch()        # this application's four weight gradients (qkv, proj, fc1, fc2) in one launch
            if self.real_token_flow:
                skip, t = norm_fork(self.norm1, x, gemm_input=True)
                # (where the crop keeps exactly the positions of the real tokens, every padded row has a zero query gradient --
                # its upstream gradient is zero -- and the fill's adjoint sums the k and v thirds only)
                fusable = self.attn.qkv.bias is not None and _WindowAttnTokensFn.applies(t, spec, C // self.attn.num_heads, self.attn.earth_position_bias_table)
                # (bf16 storage: the projection writes bf16 rows for the attention kernels, which hand bf16 rows to proj)
                lowp = fusable and _WindowAttnTokensFn.wants_bf16_qkv(B, spec, self.attn.num_heads, C // self.attn.num_heads)
                qkv_tok = self.attn.qkv(t, out_lowp=lowp, wbatch=wb)
                if fusable:
                    # partition + attention + reverse as one node whose backward is one launch (token-layout gradients)
                    t = window_attention_tokens(qkv_tok, self.attn.qkv.bias, self.attn.earth_position_bias_table, self.attn._ia, self.attn._ib,
                                                self._labels if self.roll else None, spec, fwd_shift, rev_shift, self.attn.num_heads,
                                                float(self.attn.scale), self._qrange)
                    if self.drop_path.active:        # stochastic depth: the per-sample scale rides proj's / fc2's epilogue
                        skip, t = norm_fork(self.norm2, self.drop_path.branch(self.attn.proj, t, skip, wbatch=wb), gemm_input=True)
                        return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)
                    skip, t = norm_fork(self.norm2, self.attn.proj(t, residual=skip, wbatch=wb), gemm_input=True)
                    return self.mlp(t, residual=skip, wbatch=wb)
                qkv = partition(qkv_tok, spec, fwd_shift, fill=self.attn.qkv.bias, fill_grad_from=C if self._kept_are_real else 0)
                t = self.attn.core(qkv, self._labels if self.roll else None, spec.nW, self._qrange)
                if self.drop_path.active:
                    skip, t = norm_fork(self.norm2, self.drop_path.branch(self.attn.proj, reverse(t, spec, B, rev_shift), skip, wbatch=wb), gemm_input=True)
                    return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)
                skip, t = norm_fork(self.norm2, self.attn.proj(reverse(t, spec, B, rev_shift), residual=skip, wbatch=wb), gemm_input=True)
                return self.mlp(t, residual=skip, wbatch=wb)
            skip, t = norm_fork(self.norm1, x)
            t = self.attn(partition(t, spec, fwd_shift), self._labels if self.roll else None, spec.nW)
            if self.drop_path.active:            # stochastic depth: per-sample scale fused with the residual adds
                skip, t = norm_fork(self.norm2, self.drop_path(reverse(t, spec, B, rev_shift), residual=skip), gemm_input=True)
                return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)
            skip, t = norm_fork(self.norm2, reverse(t, spec, B, rev_shift, residual=skip), gemm_input=True)
            return self.mlp(t, residual=skip, wbatch=wb)
        t = self.norm1(x).view(B, Pl, Lat, Lon, C)
        t = F.pad(t.permute(0, 4, 1, 2, 3), p).permute(0, 2, 3, 4, 1)
        Plp, Latp, Lonp = self.pad_resolution
        if self.roll:
            t = torch.roll(t, shifts=(-sh[0], -sh[1], -sh[1]), dims=(1, 2, 3))      # sic (:291)
        n_windows = (Lonp // win[2]) * (Plp // win[0]) * (Latp // win[1])
        t = self.attn(_partition(t, win), self._labels if self.roll else None, n_windows)
        t = _unpartition(t, win, B, Plp, Latp, Lonp)
        if self.roll:
            t = torch.roll(t, shifts=sh, dims=(1, 2, 3))
        t = t[:, p[4]:Plp - p[5], p[2]:Latp - p[3], p[0]:Lonp - p[1], :].reshape(B, Pl * Lat * Lon, C)
        if self.drop_path.active:
            x = self.drop_path(t, residual=x)
            return self.drop_path.branch(self.mlp, self.norm2(x), x)
        x = x + t
        return self.mlp(self.norm2(x), residual=x)


class BasicLayer(nn.Module):
    def __init__(self, dim, input_resolution, depth, num_heads, window_size, mlp_ratio=4., qkv_bias=True, qk_scale=None,
                 drop=0., attn_drop=0., drop_path=0., norm_layer=LayerNorm):
        super().__init__()
        self.blocks = nn.ModuleList([
            EarthSpecificBlock(dim, input_resolution, num_heads, window_size, (0, 0, 0) if i % 2 == 0 else None,
                               mlp_ratio, qkv_bias, qk_scale, drop, attn_drop,
                               drop_path[i] if isinstance(drop_path, list) else drop_path, norm_layer=norm_layer)
            for i in range(depth)])

    def forward(self, x):
        for blk in self.blocks:
            x = blk(x)
        return x


class DownSample(nn.Module):
    def __init__(self, in_dim, input_resolution, output_resolution):
        super().__init__()
        self.linear = Linear(in_dim * 4, in_dim * 2, bias=False)
        self.norm = LayerNorm(4 * in_dim)
        self.input_resolution, self.output_resolution = tuple(input_resolution), tuple(output_resolution)
        assert input_resolution[0] == output_resolution[0], "the dimension of pressure level shouldn't change"
        hp = output_resolution[1] * 2 - input_resolution[1]
        wp = output_resolution[2] * 2 - input_resolution[2]
        self.padding = (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2, 0, 0)

    def forward(self, x):
        B, N, C = x.shape
        (ipl, ilat, ilon), (opl, olat, olon) = self.input_resolution, self.output_resolution
        if any(self.padding):
            x = F.pad(x.reshape(B, ipl, ilat, ilon, C).permute(0, 4, 1, 2, 3), self.padding).permute(0, 2, 3, 4, 1)
        x = x.reshape(B, ipl, olat, 2, olon, 2, C).permute(0, 1, 2, 4, 3, 5, 6).reshape(B, opl * olat * olon, 4 * C)
        return self.linear(self.norm(x))


class UpSample(nn.Module):
    def __init__(self, in_dim, out_dim, input_resolution, output_resolution):
        super().__init__()
        self.linear1 = Linear(in_dim, out_dim * 4, bias=False)
        self.linear2 = Linear(out_dim, out_dim, bias=False)
        self.norm = LayerNorm(out_dim)
        self.input_resolution, self.output_resolution = tuple(input_resolution), tuple(output_resolution)

    def forward(self, x):
        B, N, C = x.shape
        (ipl, ilat, ilon), (opl, olat, olon) = self.input_resolution, self.output_resolution
        assert ipl == opl, "the dimension of pressure level shouldn't change"
        x = self.linear1(x).reshape(B, ipl, ilat, ilon, 2, 2, C // 2).permute(0, 1, 2, 4, 3, 5, 6)
        x = x.reshape(B, ipl, ilat * 2, ilon * 2, -1)
        ph, pw = ilat * 2 - olat, ilon * 2 - olon
        x = x[:, :opl, ph // 2:2 * ilat - (ph - ph // 2), pw // 2:2 * ilon - (pw - pw // 2), :]
        x = x.reshape(B, -1, x.shape[-1])
        return self.linear2(self.norm(x))


class PatchEmbed2D(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim, norm_layer=None):
        super().__init__()
        self.img_size = tuple(img_size)
        hr, wr = img_size[0] % patch_size[0], img_size[1] % patch_size[1]
        hp, wp = (patch_size[0] - hr) if hr else 0, (patch_size[1] - wr) if wr else 0
        self.padding = (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2)
        self.proj = PatchConv2d(in_chans, embed_dim, kernel_size=tuple(patch_size), stride=tuple(patch_size))
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        B, C, H, W = x.shape
        assert (H, W) == self.img_size, f"Input image size ({H}*{W}) doesn't match model ({self.img_size[0]}*{self.img_size[1]})."
        x = self.proj(F.pad(x, self.padding))
        if self.norm is not None:
            x = self.norm(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return x


class PatchRecovery2D(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, out_chans):
        super().__init__()
        self.img_size = tuple(img_size)
        self.conv = UpConvT2d(in_chans, out_chans, tuple(patch_size), tuple(patch_size))

    def forward(self, x):
        out = self.conv(x)
        H, W = out.shape[2], out.shape[3]
        ph, pw = H - self.img_size[0], W - self.img_size[1]
        return out[:, :, ph // 2:H - (ph - ph // 2), pw // 2:W - (pw - pw // 2)]

    def forward_tokens(self, tokens):
        """The same on channels-last tokens [B, Lat, Lon, C] (the layout the blocks produce): no NCHW round trip of the 2 E wide
        feature map; only the (few-channel) output is permuted."""
        out = self.conv.forward_tokens(tokens)                       # [B, H, W, O]
        H, W = out.shape[1], out.shape[2]
        ph, pw = H - self.img_size[0], W - self.img_size[1]
        return out[:, ph // 2:H - (ph - ph // 2), pw // 2:W - (pw - pw // 2)].permute(0, 3, 1, 2)


class PanguWeather(nn.Module):
    def __init__(self, constant_channels: int = 4, prescribed_channels: int = 0, prognostic_channels: int = 1,
                 embed_dim: int = 192, num_heads: tuple = (6, 12, 12, 6), window_size: tuple = (2, 6, 12),
                 patch_size: tuple = (4, 4), n_lat: int = 721, n_lon: int = 1440, context_size: int = 1, **kwargs):
        super().__init__()
        self.context_size = context_size
        window_size, patch_size = tuple(window_size), tuple(patch_size)
        drop_path = np.linspace(0, 0.2, 8).tolist()
        in_chans = constant_channels + (prescribed_channels + prognostic_channels) * context_size
        self.patchembed2d = PatchEmbed2D((n_lat, n_lon), patch_size, in_chans, embed_dim)
        res = (1, n_lat // patch_size[0], n_lon // patch_size[1])
        res2 = (1, res[1] // 2, res[2] // 2)
        self.layer1 = BasicLayer(embed_dim, res, 2, num_heads[0], window_size, drop_path=drop_path[:2])
        self.downsample = DownSample(embed_dim, res, res2)
        self.layer2 = BasicLayer(embed_dim * 2, res2, 6, num_heads[1], window_size, drop_path=drop_path[2:])
        self.layer3 = BasicLayer(embed_dim * 2, res2, 6, num_heads[2], window_size, drop_path=drop_path[2:])
        self.upsample = UpSample(embed_dim * 2, embed_dim, res2, res)
        self.layer4 = BasicLayer(embed_dim, res, 2, num_heads[3], window_size, drop_path=drop_path[:2])
        self.patchrecovery2d = PatchRecovery2D((n_lat, n_lon), patch_size, 2 * embed_dim, prognostic_channels)

    def forward_one_step(self, x):
        if getattr(self, "_drop_pool", None) is None:      # built lazily: after construction, copies and loads
            object.__setattr__(self, "_drop_pool", DropPathPool(self))
        self._drop_pool.draw(x.shape[0], x.device)        # every block's stochastic-depth mask for this call, one draw
        x = self.patchembed2d(x).unsqueeze(2)
        B, C, Pl, Lat, Lon = x.shape
        x = x.reshape(B, C, -1).transpose(1, 2)
        x = self.layer1(x)
        skip = x
        x = self.layer4(self.upsample(self.layer3(self.layer2(self.downsample(x)))))
        # reference: cat -> [B, 2E, Pl, Lat, Lon] -> level 0 -> ConvTranspose2d (panguweather.py:318-321); tokens are ordered
        # (pl, lat, lon), so level 0 is the first Lat * Lon of them and the transposed convolution reads them as they lie
        out = torch.cat([x, skip], dim=-1)
        return self.patchrecovery2d.forward_tokens(out[:, :Lat * Lon].reshape(B, Lat, Lon, -1))

    def forward(self, constants: torch.Tensor = None, prescribed: torch.Tensor = None,
                prognostic: torch.Tensor = None) -> torch.Tensor:
        return rollout(self.forward_one_step, self.context_size, constants, prescribed, prognostic)
