#!/usr/bin/env python3
"""Golden vectors for the Pangu-Weather (earth-specific 3-D window attention) path, produced by IMPORTING the
reference's own classes (/root/reference/src/dlwpbench/models/panguweather/) in this container.

    python tests/golden/make_pangu_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src/dlwpbench/models/panguweather"
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    timm = types.ModuleType("timm")
    tm = types.ModuleType("timm.models")
    layers = types.ModuleType("timm.models.layers")

    class DropPath(torch.nn.Module):
        def __init__(self, p=0.0):
            super().__init__()
            self.p = p

        def forward(self, x):
            assert not self.training, "golden vectors are taken in eval mode (stochastic depth off)"
            return x

    layers.DropPath = DropPath
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    sys.modules.update({"timm": timm, "timm.models": tm, "timm.models.layers": layers})
    # the file imports `models.panguweather.utils.*` (panguweather.py:22-27): register bare namespace packages
    for name, path in (("models", os.path.dirname(REF)), ("models.panguweather", REF),
                       ("models.panguweather.utils", os.path.join(REF, "utils"))):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    spec = importlib.util.spec_from_file_location("models.panguweather.panguweather", os.path.join(REF, "panguweather.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["models.panguweather.panguweather"] = mod
    spec.loader.exec_module(mod)
    return mod


def params(module, prefix):
    return {f"{prefix}p_{n}": p.detach().numpy() for n, p in module.named_parameters()}


def grads(module, prefix):
    return {f"{prefix}g_{n}": p.grad.numpy() for n, p in module.named_parameters() if p.grad is not None}


def main():
    ref = load_reference()
    torch.manual_seed(2468)
    out = {}
    # (i) EarthSpecificBlock, unshifted and shifted, on a (1, 10, 20) map with window (2, 6, 12): pads to (2, 12, 24)
    for tag, shift in (("plain", (0, 0, 0)), ("shift", None)):
        blk = ref.EarthSpecificBlock(dim=16, input_resolution=(1, 10, 20), num_heads=2, window_size=(2, 6, 12),
                                     shift_size=shift)
        blk.eval()
        with torch.no_grad():
            blk.attn.earth_position_bias_table.mul_(25.0)
        x = torch.randn(2, 200, 16, requires_grad=True)
        y = blk(x)
        gy = torch.randn_like(y)
        y.backward(gy)
        out.update({f"blk_{tag}_x": x.detach().numpy(), f"blk_{tag}_y": y.detach().numpy(), f"blk_{tag}_gy": gy.numpy(),
                    f"blk_{tag}_gx": x.grad.numpy()})
        out.update(params(blk, f"blk_{tag}_"))
        out.update(grads(blk, f"blk_{tag}_"))
    # (ii) whole PanguWeather, one rollout step on an 18x32 grid (pads to 20x32 for the (2,4,8) windows), embed 8
    net = ref.PanguWeather(constant_channels=2, prescribed_channels=1, prognostic_channels=3, embed_dim=8,
                           num_heads=(1, 2, 2, 1), window_size=(2, 4, 8), patch_size=(1, 1), n_lat=18, n_lon=32,
                           context_size=1)
    net.eval()
    with torch.no_grad():
        for n, p in net.named_parameters():
            if "earth_position_bias_table" in n:
                p.mul_(25.0)
    constants = torch.randn(1, 1, 2, 18, 32)
    prescribed = torch.randn(1, 2, 1, 18, 32)
    prognostic = torch.randn(1, 2, 3, 18, 32)
    target = torch.randn(1, 1, 3, 18, 32)
    y = net(constants=constants, prescribed=prescribed, prognostic=prognostic)   # T - ctx = 1 step (multi-step crashes, App. B-1)
    loss = torch.nn.functional.mse_loss(y, target)
    loss.backward()
    out.update({"net_constants": constants.numpy(), "net_prescribed": prescribed.numpy(), "net_prognostic": prognostic.numpy(),
                "net_target": target.numpy(), "net_y": y.detach().numpy(), "net_loss": np.float32(loss.item())})
    out.update(params(net, "net_"))
    out.update(grads(net, "net_"))
    path = os.path.join(OUT, "pangu_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, len(out), "arrays", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
