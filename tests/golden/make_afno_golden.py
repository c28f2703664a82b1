#!/usr/bin/env python3
"""Generate golden vectors for the AFNO path by IMPORTING the reference's own classes
(/root/reference/src/nsbench/models/fourcastnet/fourcastnet.py) in this container.

Run once here (the reference does not exist on the GPU box); the resulting .npz files are data
(seeded inputs, the reference's parameters, its outputs and gradients) and are committed.

    python tests/golden/make_afno_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/nsbench/models/fourcastnet/fourcastnet.py"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    # stubs for the two imports the file needs but this image lacks
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
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    import numpy.lib as nplib
    arraypad = types.ModuleType("numpy.lib.arraypad")
    arraypad.pad = np.pad
    sys.modules["numpy.lib.arraypad"] = arraypad
    nplib.arraypad = arraypad
    spec = importlib.util.spec_from_file_location("ref_fourcastnet", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def npz(**kw):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else v) for k, v in kw.items()}


def main():
    ref = load_reference()
    torch.manual_seed(1234)
    out = {}
    # (i) AFNO2D alone: square grid (all modes kept) and 32x64 (kept-mode quirk: 17 of 33 columns)
    for tag, (B, H, W, C, nb) in {"sq": (1, 16, 16, 64, 4), "rect": (1, 32, 64, 16, 2), "frac": (1, 16, 32, 16, 2)}.items():
        frac = 0.5 if tag == "frac" else 1.0
        m = ref.AFNO2D(C, num_blocks=nb, sparsity_threshold=0.01, hard_thresholding_fraction=frac)
        with torch.no_grad():  # larger weights than the 0.02 init so that ReLU/softshrink are exercised
            for p in m.parameters():
                p.mul_(10.0)
        x = torch.randn(B, H, W, C, requires_grad=True)
        gy = torch.randn(B, H, W, C)
        y = m(x)
        y.backward(gy)
        out.update({f"afno2d_{tag}_{k}": v for k, v in npz(
            x=x, gy=gy, y=y, gx=x.grad, w1=m.w1, b1=m.b1, w2=m.w2, b2=m.b2,
            gw1=m.w1.grad, gb1=m.b1.grad, gw2=m.w2.grad, gb2=m.b2.grad,
            meta=np.array([B, H, W, C, nb, int(frac * 100)])).items()})
    # (ii) one Block (LayerNorm eps 1e-6 as AFNONet builds it)
    from functools import partial
    blk = ref.Block(dim=32, mlp_ratio=4.0, norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_blocks=4)
    with torch.no_grad():
        for n, p in blk.named_parameters():
            if n.startswith("filter"):
                p.mul_(10.0)
    x = torch.randn(2, 8, 8, 32, requires_grad=True)
    gy = torch.randn(2, 8, 8, 32)
    y = blk(x)
    y.backward(gy)
    out.update({f"block_{k}": v for k, v in npz(x=x, gy=gy, y=y, gx=x.grad).items()})
    out.update({f"block_p_{n}": p.detach().numpy() for n, p in blk.named_parameters()})
    out.update({f"block_g_{n}": p.grad.numpy() for n, p in blk.named_parameters()})
    # (iii) AFNONet rollout: 32x32, patch 4, embed 32, depth 2, 4 blocks, context 2, T=6, tf=3
    net = ref.AFNONet(img_height=32, img_width=32, patch_size=(4, 4), in_chans=1, out_chans=1, embed_dim=32,
                      depth=2, mlp_ratio=4.0, num_blocks=4, context_size=2)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if ".filter." in n:
                p.mul_(10.0)
    x = torch.randn(2, 6, 1, 32, 32)
    target = torch.randn(2, 6, 1, 32, 32)
    y = net(x, teacher_forcing_steps=3)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out.update({f"net_{k}": v for k, v in npz(x=x, target=target, y=y, loss=loss).items()})
    out.update({f"net_p_{n}": p.detach().numpy() for n, p in net.named_parameters()})
    # note: AFNONet.norm is constructed but never used by forward_features (fourcastnet.py:251-261) -> no grad
    out.update({f"net_g_{n}": p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None})
    np.savez_compressed(os.path.join(OUT, "afno_golden.npz"), **{k: np.asarray(v, dtype=np.float32) if np.asarray(v).dtype == np.float64 else np.asarray(v) for k, v in out.items()})
    print("wrote", os.path.join(OUT, "afno_golden.npz"), len(out), "arrays")


if __name__ == "__main__":
    main()
