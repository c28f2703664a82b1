#!/usr/bin/env python3
"""Golden vectors for the window-attention path, produced by IMPORTING the reference's own classes
(/root/reference/src/nsbench/models/swintransformer/swin_transformer.py) in this container.

    python tests/golden/make_swin_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/nsbench/models/swintransformer/swin_transformer.py"
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
    spec = importlib.util.spec_from_file_location("ref_swin", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def grads(module, prefix):
    return {f"{prefix}g_{n}": p.grad.numpy() for n, p in module.named_parameters() if p.grad is not None}


def params(module, prefix):
    return {f"{prefix}p_{n}": p.detach().numpy() for n, p in module.named_parameters()}


def main():
    ref = load_reference()
    torch.manual_seed(4321)
    out = {}
    # (i) WindowAttention, window 7x7, 2 heads of dim 6, with and without the SW-MSA mask (4 windows)
    wa = ref.WindowAttention(dim=12, window_size=(7, 7), num_heads=2)
    with torch.no_grad():
        wa.relative_position_bias_table.mul_(25.0)   # make the bias matter (init std is 0.02)
    x = torch.randn(8, 49, 12, requires_grad=True)      # B_ = 2 samples x 4 windows
    labels = torch.randint(0, 3, (4, 49))
    mask = labels.unsqueeze(1) - labels.unsqueeze(2)
    mask = mask.float().masked_fill(mask != 0, -100.0)
    for tag, m in (("nomask", None), ("mask", mask)):
        wa.zero_grad()
        x.grad = None
        y = wa(x, mask=m)
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({f"wa_{tag}_x": x.detach().numpy(), f"wa_{tag}_y": y.detach().numpy(), f"wa_{tag}_gy": gy.numpy(),
                    f"wa_{tag}_gx": x.grad.numpy()})
        out.update(grads(wa, f"wa_{tag}_"))
    out.update(params(wa, "wa_"))
    out["wa_labels"] = labels.numpy().astype(np.int32)
    # (ii) BasicLayer (window 7, depth 2: W-MSA then SW-MSA) on 28x28 (multiple of 7) and 20x30 (padded)
    for tag, (H, W, pm) in {"28x28": (28, 28, "constant"), "20x30": (20, 30, "circular")}.items():
        bl = ref.BasicLayer(dim=8, depth=2, num_heads=2, window_size=7, padding_mode=pm)
        with torch.no_grad():
            for n, p in bl.named_parameters():
                if "relative_position_bias_table" in n:
                    p.mul_(25.0)
        x = torch.randn(2, H * W, 8, requires_grad=True)
        y = bl(x, H, W)[0]
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({f"bl_{tag}_x": x.detach().numpy(), f"bl_{tag}_y": y.detach().numpy(), f"bl_{tag}_gy": gy.numpy(),
                    f"bl_{tag}_gx": x.grad.numpy()})
        out.update(params(bl, f"bl_{tag}_"))
        out.update(grads(bl, f"bl_{tag}_"))
    # (iii) whole nsbench SwinTransformer (window = stage resolution, i.e. global attention with a half-map shift)
    net = ref.SwinTransformer(context_size=2, pretrain_img_size=32, patch_size=2, in_chans=1, out_chans=1, embed_dim=8,
                              depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "relative_position_bias_table" in n:
                p.mul_(25.0)
    net.eval()  # returns None in the reference (App. B-2): call as a statement
    x = torch.randn(2, 4, 1, 32, 32)
    target = torch.randn(2, 4, 1, 32, 32)
    y = net(x, teacher_forcing_steps=2)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out.update({"net_x": x.numpy(), "net_target": target.numpy(), "net_y": y.detach().numpy(),
                "net_loss": np.float32(loss.item())})
    out.update(params(net, "net_"))
    out.update(grads(net, "net_"))
    path = os.path.join(OUT, "swin_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
