#!/usr/bin/env python3
"""Golden vectors for the `ape=True` option (absolute position embedding) of both Swin models, produced by IMPORTING the
reference's classes in this container (nsbench swin_transformer.py:530-537, 640-643; dlwpbench swin_transformer.py:540-547,
650-653).  Three cases: the nsbench net on its pretraining size (the bicubic resize is the identity), the same net on a
smaller frame (24 x 24: the embedding is really resized), and the dlwpbench net over three lead times.

    python tests/golden/make_swin_ape_golden.py
"""
import os

import numpy as np
import torch

import make_dlwp_swin_golden
import make_swin_golden

OUT = os.path.dirname(os.path.abspath(__file__))


def amplify(net):
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "relative_position_bias_table" in n or n == "absolute_pos_embed":
                p.mul_(25.0)      # init std is 0.02: make both matter


def main():
    out = {}
    ns = make_swin_golden.load_reference()
    torch.manual_seed(2024)
    for tag, hw in (("ns", 32), ("ns_resized", 24)):
        net = ns.SwinTransformer(context_size=2, pretrain_img_size=32, patch_size=2, in_chans=1, out_chans=1, embed_dim=8,
                                 depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, ape=True)
        amplify(net)
        net.eval()
        x = torch.randn(2, 4, 1, hw, hw)
        target = torch.randn(2, 4, 1, hw, hw)
        y = net(x, teacher_forcing_steps=2)
        loss = torch.nn.functional.mse_loss(y, target)
        loss.backward()
        out.update({f"{tag}_x": x.numpy(), f"{tag}_target": target.numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_loss": np.float32(loss.item())})
        out.update({f"{tag}_p_{n}": p.detach().numpy() for n, p in net.named_parameters()})
        out.update({f"{tag}_g_{n}": p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None})
    dl = make_dlwp_swin_golden.load_reference()
    cfg = dict(constant_channels=2, prescribed_channels=1, prognostic_channels=2, context_size=2, img_height=16, img_width=32,
               patch_size=1, embed_dim=8, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, ape=True)
    net = dl.SwinTransformer(**cfg)
    amplify(net)
    torch.nn.Module.train(net, False)
    B, T, H, W, ctx = 2, 5, 16, 32, 2
    constants = torch.randn(B, 1, 2, H, W)
    prescribed = torch.randn(B, T, 1, H, W)
    prognostic = torch.randn(B, T, 2, H, W)
    target = torch.randn(B, T - ctx, 2, H, W)
    outs = []
    for t in range(ctx, T):      # the loop of UNet.forward around the reference's own one_step (see make_dlwp_swin_golden.py)
        prog_t = prognostic[:, t - ctx:t] if t == ctx else torch.cat(
            [prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
        x_t = net._prepare_inputs(constants=constants, prescribed=prescribed[:, t - ctx:t], prognostic=prog_t)
        outs.append(prog_t[:, -1] + net.one_step(x_t))
    y = torch.stack(outs, dim=1)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    tag = "dlwp"
    out.update({f"{tag}_constants": constants.numpy(), f"{tag}_prescribed": prescribed.numpy(),
                f"{tag}_prognostic": prognostic.numpy(), f"{tag}_target": target.numpy(), f"{tag}_y": y.detach().numpy(),
                f"{tag}_loss": np.float32(loss.item())})
    out.update({f"{tag}_p_{n}": p.detach().numpy() for n, p in net.named_parameters()})
    out.update({f"{tag}_g_{n}": p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None})
    path = os.path.join(OUT, "swin_ape_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
