"""TEST INFRASTRUCTURE ONLY — CPU oracle for the Pangu-Weather path (earth-specific 3-D window attention).
Never imported by the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

PINNED: tests/test_oracle_pangu.py checks it against tests/golden/pangu_golden.npz (reference's own classes,
tests/golden/make_pangu_golden.py).

Functional restatement of /root/reference/src/dlwpbench/models/panguweather/:
    earth_index            <- utils/earth_position_index.py:4-45
    partition3d/reverse3d  <- utils/shift_window_mask.py:4-35
    shift_labels3d         <- utils/shift_window_mask.py:38-73 (labels, not the N x N mask)
    pad3d                  <- utils/pad.py:4-34;  crop: utils/crop.py:23-43
    earth_attention        <- panguweather.py:176-211
    earth_block            <- panguweather.py:275-323 (incl. the forward roll that shifts longitude by the
                              LATITUDE shift, :291 — reproduced)
    downsample / upsample  <- panguweather.py:117-130 / 50-77
    one_step / rollout     <- panguweather.py:504-527 / 457-502 (clean on-device loop, App. B-1)
"""
import torch
import torch.nn.functional as F

DEFAULT_SHIFT = (1, 3, 6)   # panguweather.py:243: fixed, independent of the window size


def earth_index(window):
    Wpl, Wlat, Wlon = window
    z, h, w = torch.meshgrid(torch.arange(Wpl), torch.arange(Wlat), torch.arange(Wlon), indexing="ij")
    z, h, w = z.reshape(-1), h.reshape(-1), w.reshape(-1)
    s1 = 2 * Wlon - 1
    s0 = s1 * Wlat * Wlat
    return (z[:, None] + z[None, :] * Wpl) * s0 + (h[:, None] + h[None, :] * Wlat) * s1 + (w[:, None] - w[None, :] + Wlon - 1)


def partition3d(x, window):
    B, Pl, Lat, Lon, C = x.shape
    a, b, c = window
    x = x.view(B, Pl // a, a, Lat // b, b, Lon // c, c, C)
    return x.permute(0, 5, 1, 3, 2, 4, 6, 7).reshape(-1, (Pl // a) * (Lat // b), a, b, c, C)


def reverse3d(wins, window, Pl, Lat, Lon):
    a, b, c = window
    B = wins.shape[0] // (Lon // c)
    x = wins.view(B, Lon // c, Pl // a, Lat // b, a, b, c, -1)
    return x.permute(0, 2, 4, 3, 5, 1, 6, 7).reshape(B, Pl, Lat, Lon, -1)


def shift_labels3d(res, window, shift):
    Pl, Lat, Lon = res
    img = torch.zeros(1, Pl, Lat, Lon + shift[2], 1)
    cnt = 0
    for p in (slice(0, -window[0]), slice(-window[0], -shift[0]), slice(-shift[0], None)):
        for la in (slice(0, -window[1]), slice(-window[1], -shift[1]), slice(-shift[1], None)):
            for lo in (slice(0, -window[2]), slice(-window[2], -shift[2]), slice(-shift[2], None)):
                img[:, p, la, lo, :] = cnt
                cnt += 1
    lab = partition3d(img[:, :, :, :Lon, :], window)           # [n_lon, n_pl*n_lat, a, b, c, 1]
    return lab.reshape(lab.shape[0], lab.shape[1], -1).to(torch.int32)


def pad3d(res, window):
    """(left, right, top, bottom, front, back) = (lon, lon, lat, lat, pl, pl)"""
    out = []
    for n, w in ((res[2], window[2]), (res[1], window[1]), (res[0], window[0])):
        r = n % w
        p = (w - r) if r else 0
        out += [p // 2, p - p // 2]
    return tuple(out)


def earth_attention(x, p, pre, window, heads, labels=None):
    """x [B_, nW_, N, C]; labels [n_lon, nW_, N] or None."""
    B_, nW_, N, C = x.shape
    d = C // heads
    qkv = F.linear(x, p[pre + "qkv.weight"], p[pre + "qkv.bias"]).reshape(B_, nW_, N, 3, heads, d).permute(3, 0, 4, 1, 2, 5)
    q, k, v = qkv[0] * d ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)                                          # [B_, heads, nW_, N, N]
    table = p[pre + "earth_position_bias_table"]                            # [TB, types, heads]
    bias = table[earth_index(window).reshape(-1)].reshape(N, N, nW_, heads).permute(3, 2, 0, 1)
    attn = attn + bias.unsqueeze(0)
    if labels is not None:
        n_lon = labels.shape[0]
        mask = (labels[:, :, None, :] != labels[:, :, :, None]).to(x.dtype) * -100.0   # [n_lon, nW_, N, N]
        attn = (attn.view(B_ // n_lon, n_lon, heads, nW_, N, N) + mask[None, :, None]).view(-1, heads, nW_, N, N)
    attn = attn.softmax(dim=-1)
    y = (attn @ v).permute(0, 2, 3, 1, 4).reshape(B_, nW_, N, C)
    return F.linear(y, p[pre + "proj.weight"], p[pre + "proj.bias"])


def mlp(x, p, pre):
    return F.linear(F.gelu(F.linear(x, p[pre + "fc1.weight"], p[pre + "fc1.bias"])), p[pre + "fc2.weight"], p[pre + "fc2.bias"])


def earth_block(x, p, pre, res, heads, window, shift):
    Pl, Lat, Lon = res
    B, L, C = x.shape
    shortcut = x
    x = F.layer_norm(x, (C,), p[pre + "norm1.weight"], p[pre + "norm1.bias"]).view(B, Pl, Lat, Lon, C)
    pad = pad3d(res, window)
    x = F.pad(x.permute(0, 4, 1, 2, 3), pad).permute(0, 2, 3, 4, 1)
    Plp, Latp, Lonp = x.shape[1:4]
    roll = bool(shift[0] and shift[1] and shift[2])
    labels = None
    if roll:
        x = torch.roll(x, shifts=(-shift[0], -shift[1], -shift[1]), dims=(1, 2, 3))    # sic: lat shift on lon (:291)
        labels = shift_labels3d((Plp, Latp, Lonp), window, shift)
    wins = partition3d(x, window)
    wins = wins.reshape(wins.shape[0], wins.shape[1], -1, C)
    wins = earth_attention(wins, p, pre + "attn.", window, heads, labels)
    wins = wins.reshape(wins.shape[0], wins.shape[1], *window, C)
    x = reverse3d(wins, window, Plp, Latp, Lonp)
    if roll:
        x = torch.roll(x, shifts=shift, dims=(1, 2, 3))
    x = x[:, pad[4]:Plp - pad[5], pad[2]:Latp - pad[3], pad[0]:Lonp - pad[1], :].reshape(B, Pl * Lat * Lon, C)
    x = shortcut + x
    return x + mlp(F.layer_norm(x, (C,), p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "mlp.")


def basic_layer(x, p, pre, res, depth, heads, window):
    for i in range(depth):
        x = earth_block(x, p, f"{pre}blocks.{i}.", res, heads, window, (0, 0, 0) if i % 2 == 0 else DEFAULT_SHIFT)
    return x


def downsample(x, p, pre, res_in, res_out):
    B, N, C = x.shape
    x = x.reshape(B, *res_in, C)
    hp, wp = res_out[1] * 2 - res_in[1], res_out[2] * 2 - res_in[2]
    x = F.pad(x.permute(0, 4, 1, 2, 3), (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2, 0, 0)).permute(0, 2, 3, 4, 1)
    x = x.reshape(B, res_in[0], res_out[1], 2, res_out[2], 2, C).permute(0, 1, 2, 4, 3, 5, 6)
    x = x.reshape(B, res_out[0] * res_out[1] * res_out[2], 4 * C)
    x = F.layer_norm(x, (4 * C,), p[pre + "norm.weight"], p[pre + "norm.bias"])
    return F.linear(x, p[pre + "linear.weight"])


def upsample(x, p, pre, res_in, res_out):
    B, N, C = x.shape
    x = F.linear(x, p[pre + "linear1.weight"])
    x = x.reshape(B, res_in[0], res_in[1], res_in[2], 2, 2, C // 2).permute(0, 1, 2, 4, 3, 5, 6)
    x = x.reshape(B, res_in[0], res_in[1] * 2, res_in[2] * 2, -1)
    ph, pw = res_in[1] * 2 - res_out[1], res_in[2] * 2 - res_out[2]
    x = x[:, :res_out[0], ph // 2:2 * res_in[1] - (ph - ph // 2), pw // 2:2 * res_in[2] - (pw - pw // 2), :]
    x = x.reshape(B, -1, x.shape[-1])
    x = F.layer_norm(x, (x.shape[-1],), p[pre + "norm.weight"], p[pre + "norm.bias"])
    return F.linear(x, p[pre + "linear2.weight"])


def one_step(x, p, cfg):
    E, heads, window, ps = cfg["embed_dim"], cfg["num_heads"], tuple(cfg["window_size"]), cfg["patch_size"]
    H, W = cfg["n_lat"], cfg["n_lon"]
    hr, wr = H % ps[0], W % ps[1]
    hp, wp = (ps[0] - hr) if hr else 0, (ps[1] - wr) if wr else 0
    x = F.pad(x, (wp // 2, wp - wp // 2, hp // 2, hp - hp // 2))
    x = F.conv2d(x, p["patchembed2d.proj.weight"], p["patchembed2d.proj.bias"], stride=ps).unsqueeze(2)
    B, C, Pl, Lat, Lon = x.shape
    res = (1, H // ps[0], W // ps[1])
    res2 = (1, res[1] // 2, res[2] // 2)
    x = x.reshape(B, C, -1).transpose(1, 2)
    x = basic_layer(x, p, "layer1.", res, 2, heads[0], window)
    skip = x
    x = downsample(x, p, "downsample.", res, res2)
    x = basic_layer(x, p, "layer2.", res2, 6, heads[1], window)
    x = basic_layer(x, p, "layer3.", res2, 6, heads[2], window)
    x = upsample(x, p, "upsample.", res2, res)
    x = basic_layer(x, p, "layer4.", res, 2, heads[3], window)
    out = torch.cat([x, skip], dim=-1).transpose(1, 2).reshape(B, -1, Pl, Lat, Lon)[:, :, 0]
    out = F.conv_transpose2d(out, p["patchrecovery2d.conv.weight"], p["patchrecovery2d.conv.bias"], stride=ps)
    Ho, Wo = out.shape[2], out.shape[3]
    ph, pw = Ho - H, Wo - W
    return out[:, :, ph // 2:Ho - (ph - ph // 2), pw // 2:Wo - (pw - pw // 2)]


def rollout(constants, prescribed, prognostic, p, cfg):
    ctx = cfg["context_size"]
    outs = []
    for t in range(ctx, prognostic.shape[1]):
        prog_t = prognostic[:, max(0, t - ctx):t] if t == ctx else torch.cat(
            [prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        parts = []
        if constants is not None:
            parts.append(constants[:, 0])
        if prescribed is not None:
            parts.append(prescribed[:, t - ctx:t].flatten(1, 2))
        parts.append(prog_t.flatten(1, 2))
        outs.append(prog_t[:, -1] + one_step(torch.cat(parts, dim=1), p, cfg))
    return torch.stack(outs, dim=1)
