#!/usr/bin/env python3
"""Golden vectors for the dlwpbench SwinTransformer, produced by IMPORTING the reference's class
(/root/reference/src/dlwpbench/models/swintransformer/swin_transformer.py) in this container.

(one) a single lead time through the reference's own forward(); (multi) three lead times with context 2, where the
reference's forward() raises at the second lead time: the loop of UNet.forward (unet.py:64-111) is
driven by hand around the reference's own one_step().

    python tests/golden/make_dlwp_swin_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/dlwpbench/models/swintransformer/swin_transformer.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert not self.training or self.p == 0.0
            return x

    layers.DropPath = DropPath
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    layers.to_2tuple = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    spec = importlib.util.spec_from_file_location("ref_dlwp_swin", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    torch.manual_seed(888)
    out = {}
    cfgs = {"one": dict(constant_channels=2, prescribed_channels=1, prognostic_channels=3, context_size=1, img_height=16,
                        img_width=32, patch_size=2, embed_dim=8, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, T=2),
            "multi": dict(constant_channels=2, prescribed_channels=1, prognostic_channels=2, context_size=2, img_height=16,
                          img_width=32, patch_size=1, embed_dim=8, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, T=5)}
    for tag, cfg in cfgs.items():
        T = cfg.pop("T")
        net = ref.SwinTransformer(**cfg)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if "relative_position_bias_table" in n:
                    p.mul_(25.0)
        torch.nn.Module.train(net, False)   # the class overrides train() (:738-741)
        B, H, W, ctx = 2, cfg["img_height"], cfg["img_width"], cfg["context_size"]
        constants = torch.randn(B, 1, cfg["constant_channels"], H, W)
        prescribed = torch.randn(B, T, cfg["prescribed_channels"], H, W)
        prognostic = torch.randn(B, T, cfg["prognostic_channels"], H, W)
        target = torch.randn(B, T - ctx, cfg["prognostic_channels"], H, W)
        if tag == "one":
            y = net(constants=constants, prescribed=prescribed, prognostic=prognostic)
        else:
            outs = []
            for t in range(ctx, T):
                prog_t = prognostic[:, t - ctx:t] if t == ctx else torch.cat(
                    [prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
                x_t = net._prepare_inputs(constants=constants, prescribed=prescribed[:, t - ctx:t], prognostic=prog_t)
                outs.append(prog_t[:, -1] + net.one_step(x_t))
            y = torch.stack(outs, dim=1)
        loss = torch.nn.functional.mse_loss(y, target)
        loss.backward()
        out.update({f"{tag}_constants": constants.numpy(), f"{tag}_prescribed": prescribed.numpy(),
                    f"{tag}_prognostic": prognostic.numpy(), f"{tag}_target": target.numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_loss": np.float32(loss.item())})
        out.update({f"{tag}_p_{n}": p.detach().numpy() for n, p in net.named_parameters()})
        out.update({f"{tag}_g_{n}": p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None})
    path = os.path.join(OUT, "dlwp_swin_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
