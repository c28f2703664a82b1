"""TEST INFRASTRUCTURE ONLY — CPU oracle for the window-attention (Swin) path.  Never imported by the
product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

PINNED: tests/test_oracle_swin.py checks it against tests/golden/swin_golden.npz, produced by
tests/golden/make_swin_golden.py from the reference's own classes.

Functional restatement (parameter dicts, no nn.Module) of
/root/reference/src/nsbench/models/swintransformer/swin_transformer.py:
    rel_index / window_attention  <- WindowAttention            :75-155
    shift_labels                  <- BasicLayer.forward mask     :377-395 (labels, not the N x N mask)
    swin_block                    <- SwinTransformerBlock.forward :201-258
    patch_merging                 <- PatchMerging.forward        :275-302
    basic_layer                   <- BasicLayer.forward          :368-408
    swin_one_step / swin_rollout  <- SwinTransformer.one_step / .forward :635-700
"""
import math

import torch
import torch.nn.functional as F


def rel_index(Wh, Ww):
    """relative_position_index [N,N] for a Wh x Ww window (:102-112), N = Wh*Ww row-major."""
    ys, xs = torch.meshgrid(torch.arange(Wh), torch.arange(Ww), indexing="ij")
    ys, xs = ys.reshape(-1), xs.reshape(-1)
    dy = ys[:, None] - ys[None, :] + Wh - 1
    dx = xs[:, None] - xs[None, :] + Ww - 1
    return dy * (2 * Ww - 1) + dx


def window_attention(x, p, pre, Wh, Ww, heads, labels=None):
    """x [B_,N,C]; labels [nW,N] ints or None (mask = -100 where labels differ)."""
    B_, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"]).reshape(B_, N, 3, heads, d).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    bias = p[pre + "relative_position_bias_table"][rel_index(Wh, Ww).reshape(-1)].reshape(N, N, heads).permute(2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if labels is not None:
        nW = labels.shape[0]
        mask = (labels[:, None, :] != labels[:, :, None]).to(x.dtype) * -100.0        # [nW,N,N]
        attn = (attn.view(B_ // nW, nW, heads, N, N) + mask[None, :, None]).view(-1, heads, N, N)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).transpose(1, 2).reshape(B_, N, C)
    return F.linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"])


def partition(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)


def unpartition(wins, ws, H, W):
    B = wins.shape[0] // ((H // ws) * (W // ws))
    return wins.view(B, H // ws, W // ws, ws, ws, -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_labels(Hp, Wp, ws, shift):
    """Region labels of the shifted canvas, per window: [nW, ws*ws] (BasicLayer.forward :380-393)."""
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for hs in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
        for wsl in (slice(0, -ws), slice(-ws, -shift), slice(-shift, None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    return partition(img, ws).reshape(-1, ws * ws).to(torch.int32)


def mlp(x, p, pre):
    return F.linear(F.gelu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])), p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def pad_hw(x, pad_b, pad_r, padding_mode):
    """x [B,H,W,C] padded at the bottom / right (SwinTransformerBlock.forward :219-221).  padding_mode: one mode for both axes
    (the reference's argument) or a (latitude, longitude) pair -- the dlwpbench intent "constant latitude, circular longitude",
    which the dlwpbench block itself applies to the wrong axes (:218-222)."""
    if isinstance(padding_mode, str):
        return F.pad(x, (0, 0, 0, pad_r, 0, pad_b), mode=padding_mode)
    if pad_r:
        x = F.pad(x, (0, 0, 0, pad_r, 0, 0), mode=padding_mode[1])
    if pad_b:
        x = F.pad(x, (0, 0, 0, 0, 0, pad_b), mode=padding_mode[0])
    return x


def swin_block(x, p, pre, H, W, ws, shift, heads, labels, padding_mode="constant"):
    B, L, C = x.shape
    shortcut = x
    x = F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"]).view(B, H, W, C)
    pad_r, pad_b = (ws - W % ws) % ws, (ws - H % ws) % ws
    x = pad_hw(x, pad_b, pad_r, padding_mode)
    Hp, Wp = x.shape[1], x.shape[2]
    if shift > 0:
        x = torch.roll(x, shifts=(-shift, -shift), dims=(1, 2))
    wins = window_attention(partition(x, ws), p, pre + "attn.", ws, ws, heads, labels if shift > 0 else None)
    x = unpartition(wins, ws, Hp, Wp)
    if shift > 0:
        x = torch.roll(x, shifts=(shift, shift), dims=(1, 2))
    x = x[:, :H, :W, :].reshape(B, H * W, C)
    x = shortcut + x
    return x + mlp(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")


def patch_merging(x, p, pre, H, W, padding_mode="constant"):
    B, L, C = x.shape
    x = x.view(B, H, W, C)
    if H % 2 or W % 2:
        x = pad_hw(x, H % 2, W % 2, padding_mode)
    x = torch.cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1).reshape(B, -1, 4 * C)
    x = F.layer_norm(x, (4 * C,), p[pre + "norm.weight"], p[pre + "norm.bias"])
    return F.linear(x, p[pre + "reduction.weight"])


def basic_layer(x, p, pre, H, W, ws, depth, heads, downsample, padding_mode="constant"):
    Hp, Wp = math.ceil(H / ws) * ws, math.ceil(W / ws) * ws
    labels = shift_labels(Hp, Wp, ws, ws // 2)
    for i in range(depth):
        x = swin_block(x, p, f"{pre}blocks.{i}.", H, W, ws, 0 if i % 2 == 0 else ws // 2, heads, labels, padding_mode)
    if downsample:
        return x, H, W, patch_merging(x, p, pre + "downsample.", H, W, padding_mode), (H + 1) // 2, (W + 1) // 2
    return x, H, W, x, H, W


def add_ape(x, p, Wh, Ww):
    """`ape=True` (nsbench :640-643, dlwpbench :650-653): the learned [1, E, Wh0, Ww0] embedding, resized bicubically to the
    token grid, added to the embedded patches (after the patch norm).  Present in `p` only when the model was built with it."""
    if "absolute_pos_embed" not in p:
        return x
    return x + F.interpolate(p["absolute_pos_embed"], size=(Wh, Ww), mode="bicubic").flatten(2).transpose(1, 2)


def swin_one_step(x, p, cfg):
    ps, E, depths, heads = cfg["patch_size"], cfg["embed_dim"], cfg["depths"], cfg["num_heads"]
    x = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=ps)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    if cfg.get("patch_norm", True):
        x = F.layer_norm(x, (E,), p["patch_embed.norm.weight"], p["patch_embed.norm.bias"])
    x = add_ape(x, p, Wh, Ww)
    res = cfg["pretrain_img_size"] // ps
    outs = []
    for i in range(len(depths)):
        x_out, H, W, x, Wh, Ww = basic_layer(x, p, f"layers.{i}.", Wh, Ww, res, depths[i], heads[i], i < len(depths) - 1)
        C = E * 2 ** i
        x_out = F.layer_norm(x_out, (C,), p[f"norm{i}.weight"], p[f"norm{i}.bias"])
        outs.append(x_out.view(-1, H, W, C).permute(0, 3, 1, 2))
        res //= 2
    outs = outs[::-1]
    x_out = None
    for idx in range(len(depths)):
        x_in = outs[idx] if idx == 0 else torch.cat([outs[idx], x_out], dim=1)
        x_out = F.gelu(F.conv_transpose2d(x_in, p[f"decoder.{idx}.0.weight"], p[f"decoder.{idx}.0.bias"], stride=2))
    return F.conv2d(x_out, p["final.weight"], p["final.bias"])


def swin_rollout(x, p, cfg, teacher_forcing_steps):
    """SwinTransformer.forward (:669-700): same windowing as the other ns models, residual output (:696)."""
    ctx = cfg["context_size"]
    outs, out = [], None
    for t in range(x.shape[1]):
        if t < teacher_forcing_steps:
            x_t = x[:, max(0, t - (ctx - 1)):t + 1]
        else:
            ts = max(0, (teacher_forcing_steps - t - 1) + ctx)
            x_t = torch.cat([x[:, teacher_forcing_steps - ts:teacher_forcing_steps],
                             torch.stack(outs[-(ctx - ts):], dim=1)], dim=1)
        out = x_t[:, -1] if t < ctx - 1 else x_t[:, -1] + swin_one_step(x_t.flatten(1, 2), p, cfg)
        outs.append(out)
    return torch.stack(outs, dim=1)


# ---- dlwpbench twin (src/dlwpbench/models/swintransformer/swin_transformer.py) ----------------------------------
# Same layers with (h, w) window pairs (:52-69, :383-399); every stage's window is its whole feature map
# (resolution = (H/p, W/p), halved per stage, :542-571), so no window padding is ever applied -- with smaller windows
# the reference's pads land on the wrong axes (:218-222) and BasicLayer raises.
def partition2(x, ws):
    B, H, W, C = x.shape
    return x.view(B, H // ws[0], ws[0], W // ws[1], ws[1], C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws[0] * ws[1], C)


def unpartition2(wins, ws, H, W):
    B = wins.shape[0] // ((H // ws[0]) * (W // ws[1]))
    return wins.view(B, H // ws[0], W // ws[1], ws[0], ws[1], -1).permute(0, 1, 3, 2, 4, 5).reshape(B, H, W, -1)


def shift_labels2(Hp, Wp, ws, shift):
    img = torch.zeros(1, Hp, Wp, 1)
    cnt = 0
    for hs in (slice(0, -ws[0]), slice(-ws[0], -shift[0]), slice(-shift[0], None)):
        for wsl in (slice(0, -ws[1]), slice(-ws[1], -shift[1]), slice(-shift[1], None)):
            img[:, hs, wsl, :] = cnt
            cnt += 1
    return partition2(img, ws).reshape(-1, ws[0] * ws[1]).to(torch.int32)


def dlwp_swin_block(x, p, pre, H, W, ws, shift, heads, labels):
    B, L, C = x.shape
    assert H % ws[0] == 0 and W % ws[1] == 0, "dlwpbench windows are whole feature maps"
    shortcut = x
    x = F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"]).view(B, H, W, C)
    shifted = shift[0] > 0 or shift[1] > 0
    if shifted:
        x = torch.roll(x, shifts=(-shift[0], -shift[1]), dims=(1, 2))
    wins = window_attention(partition2(x, ws), p, pre + "attn.", ws[0], ws[1], heads, labels if shifted else None)
    x = unpartition2(wins, ws, H, W)
    if shifted:
        x = torch.roll(x, shifts=(shift[0], shift[1]), dims=(1, 2))
    x = shortcut + x.reshape(B, H * W, C)
    return x + mlp(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")


def dlwp_basic_layer(x, p, pre, H, W, ws, depth, heads, downsample):
    shift = (ws[0] // 2, ws[1] // 2)
    labels = shift_labels2(H, W, ws, shift)
    for i in range(depth):
        x = dlwp_swin_block(x, p, f"{pre}blocks.{i}.", H, W, ws, (0, 0) if i % 2 == 0 else shift, heads, labels)
    if downsample:
        return x, H, W, patch_merging(x, p, pre + "downsample.", H, W), (H + 1) // 2, (W + 1) // 2
    return x, H, W, x, H, W


def dlwp_swin_one_step(x, p, cfg):
    """SwinTransformer.one_step (:645-677) with the dlwpbench constructor (:494-608): decoder stage 0 (the last
    transposed convolution) has kernel = stride = patch_size, the others 2 (:593-594).
    cfg["window_size"] (an int; NOT a reference key: the reference fixes window = stage resolution, :542,561): classic Swin
    windows through the nsbench BasicLayer (pinned at window 7 on 32 x 64 / 128 x 256 by c4_window7_golden.npz) with constant
    latitude / circular longitude padding -- BASELINE configs[3] ("window=7"), which the dlwpbench block cannot run (App. B-6)."""
    ps, E, depths, heads = cfg["patch_size"], cfg["embed_dim"], cfg["depths"], cfg["num_heads"]
    x = F.conv2d(x, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=ps)
    Wh, Ww = x.shape[2], x.shape[3]
    x = x.flatten(2).transpose(1, 2)
    if cfg.get("patch_norm", True):
        x = F.layer_norm(x, (E,), p["patch_embed.norm.weight"], p["patch_embed.norm.bias"])
    x = add_ape(x, p, Wh, Ww)
    res = (cfg["img_height"] // ps, cfg["img_width"] // ps)
    outs = []
    for i in range(len(depths)):
        if cfg.get("window_size"):
            x_out, H, W, x, Wh, Ww = basic_layer(x, p, f"layers.{i}.", Wh, Ww, int(cfg["window_size"]), depths[i], heads[i],
                                                 i < len(depths) - 1, ("constant", "circular"))
        else:
            x_out, H, W, x, Wh, Ww = dlwp_basic_layer(x, p, f"layers.{i}.", Wh, Ww, res, depths[i], heads[i], i < len(depths) - 1)
        C = E * 2 ** i
        x_out = F.layer_norm(x_out, (C,), p[f"norm{i}.weight"], p[f"norm{i}.bias"])
        outs.append(x_out.view(-1, H, W, C).permute(0, 3, 1, 2))
        res = (res[0] // 2, res[1] // 2)
    outs = outs[::-1]
    x_out = None
    for idx in range(len(depths)):
        x_in = outs[idx] if idx == 0 else torch.cat([outs[idx], x_out], dim=1)
        stride = ps if idx == len(depths) - 1 else 2
        x_out = F.gelu(F.conv_transpose2d(x_in, p[f"decoder.{idx}.0.weight"], p[f"decoder.{idx}.0.bias"], stride=stride))
    return F.conv2d(x_out, p["final.weight"], p["final.bias"])


def dlwp_swin(constants, prescribed, prognostic, p, cfg):
    """The dlwpbench rollout in its working form (UNet.forward unet.py:64-111; the copy at :694-737 raises at the
    second lead time): out_t = prog_t[:, -1] + one_step(x_t)."""
    ctx, outs = cfg["context_size"], []
    for t in range(ctx, prognostic.shape[1]):
        if t == ctx:
            prog_t = prognostic[:, max(0, t - ctx):t]
        else:
            prog_t = torch.cat([prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        parts = [] if constants is None else [constants[:, 0]]
        if prescribed is not None:
            parts.append(prescribed[:, t - ctx:t].flatten(1, 2))
        parts.append(prog_t.flatten(1, 2))
        outs.append(prog_t[:, -1] + dlwp_swin_one_step(torch.cat(parts, dim=1), p, cfg))
    return torch.stack(outs, dim=1)
