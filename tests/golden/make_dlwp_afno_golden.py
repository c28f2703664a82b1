#!/usr/bin/env python3
"""Golden vectors for the dlwpbench AFNONet (FourCastNet), produced by IMPORTING the reference's class
(/root/reference/src/dlwpbench/models/fourcastnet/fourcastnet.py) in this container.

Two cases: (one) a single lead time through the reference's own forward(); (multi) three lead times with context 2,
where the reference's forward() raises at the second lead time (`list.to()`) -- there the loop of
UNet.forward (unet.py:64-111) is driven by hand around the reference's own forward_features / head.

    python tests/golden/make_dlwp_afno_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/dlwpbench/models/fourcastnet/fourcastnet.py"
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
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.layers": layers})
    import numpy.lib as nplib
    arraypad = types.ModuleType("numpy.lib.arraypad")
    arraypad.pad = np.pad
    sys.modules["numpy.lib.arraypad"] = arraypad
    nplib.arraypad = arraypad
    # third-party modules the file imports but AFNONet never uses
    for name, attrs in {"neuralop": (), "neuralop.models": ("FNO",), "torch_harmonics": (), "torch_harmonics.examples": (),
                        "torch_harmonics.examples.sfno": ("SphericalFourierNeuralOperatorNet",)}.items():
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, object)
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("ref_dlwp_fourcastnet", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    torch.manual_seed(777)
    out = {}
    cfgs = {"one": dict(img_height=16, img_width=32, patch_size=(2, 2), constant_channels=2, prescribed_channels=1,
                        prognostic_channels=3, embed_dim=32, depth=2, mlp_ratio=2.0, num_blocks=4, context_size=1, T=2),
            "multi": dict(img_height=16, img_width=32, patch_size=(4, 4), constant_channels=2, prescribed_channels=1,
                          prognostic_channels=2, embed_dim=32, depth=2, mlp_ratio=2.0, num_blocks=4, context_size=2, T=5,
                          use_pos_embed=False)}
    for tag, cfg in cfgs.items():
        T = cfg.pop("T")
        net = ref.AFNONet(**cfg)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if ".filter." in n:
                    p.mul_(10.0)
        B, H, W = 2, cfg["img_height"], cfg["img_width"]
        constants = torch.randn(B, 1, cfg["constant_channels"], H, W)
        prescribed = torch.randn(B, T, cfg["prescribed_channels"], H, W)
        prognostic = torch.randn(B, T, cfg["prognostic_channels"], H, W)
        ctx = cfg["context_size"]
        target = torch.randn(B, T - ctx, cfg["prognostic_channels"], H, W)
        if tag == "one":
            y = net(constants=constants, prescribed=prescribed, prognostic=prognostic)
        else:
            ph, pw = cfg["patch_size"]
            outs = []
            for t in range(ctx, T):
                prog_t = prognostic[:, t - ctx:t] if t == ctx else torch.cat(
                    [prognostic[:, max(0, t - ctx):ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
                x_t = net._prepare_inputs(constants=constants, prescribed=prescribed[:, t - ctx:t], prognostic=prog_t)
                z = net.head(net.forward_features(x_t))
                z = z.reshape(B, H // ph, W // pw, ph, pw, -1).permute(0, 5, 1, 3, 2, 4).reshape(B, -1, H, W)
                outs.append(prog_t[:, -1] + z)
            y = torch.stack(outs, dim=1)
        loss = torch.nn.functional.mse_loss(y, target)
        loss.backward()
        out.update({f"{tag}_constants": constants.numpy(), f"{tag}_prescribed": prescribed.numpy(),
                    f"{tag}_prognostic": prognostic.numpy(), f"{tag}_target": target.numpy(), f"{tag}_y": y.detach().numpy(),
                    f"{tag}_loss": np.float32(loss.item())})
        out.update({f"{tag}_p_{n}": p.detach().numpy() for n, p in net.named_parameters()})
        out.update({f"{tag}_g_{n}": p.grad.numpy() for n, p in net.named_parameters() if p.grad is not None})
    path = os.path.join(OUT, "dlwp_afno_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
