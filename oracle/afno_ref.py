"""TEST INFRASTRUCTURE ONLY — CPU oracle for the AFNO (FourCastNet) path.  Never imported by the
product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

PINNED: checked by tests/test_oracle_afno.py against tests/golden/afno_golden.npz, which
tests/golden/make_afno_golden.py produced by importing the reference's own classes.

Restated from /root/reference/src/nsbench/models/fourcastnet/fourcastnet.py:
    afno2d        <- AFNO2D.forward   :77-126   (kept-mode window computed from H only, :92-93)
    block         <- Block.forward    :152-165  (LayerNorm eps 1e-6 set at :213; double skip)
    mlp           <- Mlp.forward      :50-56
    afnonet_step  <- forward_features :251-261 + head / un-patchify / residual :286-296
    afnonet       <- AFNONet.forward  :263-300  (same windowing as the FNO rollout)
The complex block-diagonal MLP is written with complex tensors here (the reference spells it as
four real einsums); ReLU and soft-shrink act on real and imaginary parts independently.
"""
import torch
import torch.nn.functional as F


def kept_window(H, W, frac):
    total = H // 2 + 1
    kept = int(total * frac)
    r0, r1 = max(0, total - kept), min(H, total + kept)
    c1 = min(W // 2 + 1, kept)
    return r0, r1, c1


def afno2d(x, w1, b1, w2, b2, num_blocks, sparsity_threshold=0.01, hard_thresholding_fraction=1.0):
    """x [B,H,W,C]; w1 [2,nb,bs,bs*f]; b1 [2,nb,bs*f]; w2 [2,nb,bs*f,bs]; b2 [2,nb,bs]."""
    B, H, W, C = x.shape
    nb = num_blocks
    bs = C // nb
    X = torch.fft.rfft2(x.float(), dim=(1, 2), norm="ortho").reshape(B, H, W // 2 + 1, nb, bs)
    r0, r1, c1 = kept_window(H, W, hard_thresholding_fraction)
    W1 = torch.complex(w1[0], w1[1])
    W2 = torch.complex(w2[0], w2[1])
    Xk = X[:, r0:r1, :c1]
    o1 = torch.einsum("...bi,bio->...bo", Xk, W1)
    o1 = torch.complex(F.relu(o1.real + b1[0]), F.relu(o1.imag + b1[1]))
    o2 = torch.einsum("...bi,bio->...bo", o1, W2)
    o2 = torch.complex(o2.real + b2[0], o2.imag + b2[1])
    full = torch.zeros_like(X)
    full[:, r0:r1, :c1] = o2
    full = torch.view_as_complex(F.softshrink(torch.view_as_real(full), lambd=sparsity_threshold))
    y = torch.fft.irfft2(full.reshape(B, H, W // 2 + 1, C), s=(H, W), dim=(1, 2), norm="ortho")
    return y.type(x.dtype) + x


def mlp(x, p, prefix):
    h = F.gelu(F.linear(x, p[prefix + "fc1.weight"], p[prefix + "fc1.bias"]))
    return F.linear(h, p[prefix + "fc2.weight"], p[prefix + "fc2.bias"])


def block(x, p, prefix, num_blocks, sparsity_threshold=0.01, hard_thresholding_fraction=1.0, eps=1e-6):
    C = x.shape[-1]
    residual = x
    x = F.layer_norm(x, (C,), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"], eps)
    x = afno2d(x, p[prefix + "filter.w1"], p[prefix + "filter.b1"], p[prefix + "filter.w2"], p[prefix + "filter.b2"],
               num_blocks, sparsity_threshold, hard_thresholding_fraction)
    x = x + residual
    residual = x
    x = F.layer_norm(x, (C,), p[prefix + "norm2.weight"], p[prefix + "norm2.bias"], eps)
    x = mlp(x, p, prefix + "mlp.")
    return x + residual


def afnonet_step(x_in, p, cfg):
    """x_in [B, ctx*D, H, W] -> network increment [B, out_chans, H, W] (before the residual)."""
    ph, pw = cfg["patch_size"]
    B = x_in.shape[0]
    h, w = cfg["img_height"] // ph, cfg["img_width"] // pw
    t = F.conv2d(x_in, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=(ph, pw))
    t = t.flatten(2).transpose(1, 2) + p["pos_embed"]
    t = t.reshape(B, h, w, cfg["embed_dim"])
    for i in range(cfg["depth"]):
        t = block(t, p, f"blocks.{i}.", cfg["num_blocks"], cfg.get("sparsity_threshold", 0.01),
                  cfg.get("hard_thresholding_fraction", 1.0))
    t = F.linear(t, p["head.weight"])  # [B,h,w,p1*p2*c_out]
    co = cfg["out_chans"]
    t = t.reshape(B, h, w, ph, pw, co).permute(0, 5, 1, 3, 2, 4).reshape(B, co, h * ph, w * pw)
    return t


def afnonet(x, p, cfg, teacher_forcing_steps):
    """AFNONet.forward: x [B,T,D,H,W] -> [B,T,D,H,W]"""
    ctx = cfg["context_size"]
    outs, out = [], None
    for t in range(x.shape[1]):
        if t < teacher_forcing_steps:
            x_t_in = x[:, max(0, t - (ctx - 1)):t + 1]
        else:
            if ctx == 0:
                x_t_in = out
            else:
                ts = max(0, (teacher_forcing_steps - t - 1) + ctx)
                x_obs = x[:, teacher_forcing_steps - ts:teacher_forcing_steps]
                x_out = torch.stack(outs[-(ctx - ts):], dim=1)
                x_t_in = torch.cat([x_obs, x_out], dim=1)
        if t < ctx - 1:
            out = x_t_in[:, -1]
        else:
            out = x_t_in[:, -1] + afnonet_step(x_t_in.flatten(1, 2), p, cfg)
        outs.append(out)
    return torch.stack(outs, dim=1)


# ---- dlwpbench twin (src/dlwpbench/models/fourcastnet/fourcastnet.py:215-361) ------------------------------
def dlwp_prepare_inputs(constants, prescribed, prognostic):
    """_prepare_inputs :299-311: constants[:,0] | prescribed "b (t c) h w" | prognostic "b (t c) h w"."""
    parts = [] if constants is None else [constants[:, 0]]
    if prescribed is not None:
        parts.append(prescribed.flatten(1, 2))
    if prognostic is not None:
        parts.append(prognostic.flatten(1, 2))
    return torch.cat(parts, dim=1)


def dlwp_rollout(one_step, ctx, constants, prescribed, prognostic):
    """The dlwpbench loop in its working form (UNet.forward unet.py:64-111; the AFNONet copy :313-361 calls
    `.to()` on a list at the second lead time): out_t = prog_t[:, -1] + net(x_t), t = ctx .. T-1."""
    outs = []
    for t in range(ctx, prognostic.shape[1]):
        if t == ctx:
            prog_t = prognostic[:, max(0, t - ctx):t]
        else:
            prog_t = torch.cat([prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        x_t = dlwp_prepare_inputs(constants, prescribed[:, t - ctx:t] if prescribed is not None else None, prog_t)
        outs.append(prog_t[:, -1] + one_step(x_t))
    return torch.stack(outs, dim=1)


def dlwp_afnonet_step(x_in, p, cfg):
    """forward_features :287-297 (+pos_embed only if use_pos_embed) + head + un-patchify :344-357."""
    ph, pw = cfg["patch_size"]
    B = x_in.shape[0]
    h, w = cfg["img_height"] // ph, cfg["img_width"] // pw
    t = F.conv2d(x_in, p["patch_embed.proj.weight"], p["patch_embed.proj.bias"], stride=(ph, pw))
    t = t.flatten(2).transpose(1, 2)
    if "pos_embed" in p:
        t = t + p["pos_embed"]
    t = t.reshape(B, h, w, cfg["embed_dim"])
    for i in range(cfg["depth"]):
        t = block(t, p, f"blocks.{i}.", cfg["num_blocks"], cfg.get("sparsity_threshold", 0.01),
                  cfg.get("hard_thresholding_fraction", 1.0))
    t = F.linear(t, p["head.weight"])
    co = cfg["prognostic_channels"]
    return t.reshape(B, h, w, ph, pw, co).permute(0, 5, 1, 3, 2, 4).reshape(B, co, h * ph, w * pw)


def dlwp_afnonet(constants, prescribed, prognostic, p, cfg):
    return dlwp_rollout(lambda x: dlwp_afnonet_step(x, p, cfg), cfg["context_size"], constants, prescribed, prognostic)
