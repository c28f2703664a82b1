#!/usr/bin/env python3
"""Golden vectors that pin the IN-TREE FNO rollout drivers by executing the reference's own classes.

The arithmetic of the FNO backbone lives in the third-party package `neuralop` (absent here; oracle/fno_ref.py restates
it, PARITY UNPINNED).  The rollout drivers around it are the reference's own code:
    nsbench   TFNO2DModule.forward  src/nsbench/models/fno/fno.py:217-250   (context window, teacher forcing, closed loop)
    nsbench   FNOModule.forward     src/nsbench/models/fno/fno.py:29-41
    dlwpbench FNO2DModule._prepare_inputs / forward   src/dlwpbench/models/fno/fno.py:49-62, 64-106
This script imports those classes by path with a stub `neuralop.models` whose FNO / TFNO is the seeded network of
oracle/fno_ref.py wrapped as an nn.Module, runs them on seeded inputs and stores inputs, parameters, outputs, loss and all
gradients.  Whatever the backbone computes, the windowing, channel order, residual and BPTT structure in the vectors are
the reference's.  dlwpbench's published multi-step loop raises (list.to(), :91-95): its single-lead-time
forward is executed as published; the multi-step vectors drive the reference's own `_prepare_inputs` and `self.fno` with
the clean loop (UNet.forward, dlwpbench/models/unet/unet.py:64-111), like make_dlwp_afno_golden.py does for AFNONet.

    python tests/golden/make_fno_driver_golden.py        (writes tests/golden/fno_driver_golden.npz)
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fno_ref  # noqa: E402

REF = "/root/reference/src"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fno_driver_golden.npz")
_SEED = [100]


class StubFNO(nn.Module):
    """neuralop.models.FNO stand-in: oracle/fno_ref.FNO's arithmetic with nn.Parameters (seeded per construction)."""

    def __init__(self, n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels, n_layers,
                 max_n_modes=None, rank=1.0, **kwargs):
        super().__init__()
        _SEED[0] += 1
        self.core = fno_ref.FNO(n_modes, in_channels, hidden_channels, lifting_channels, projection_channels, out_channels,
                                n_layers, seed=_SEED[0])
        self.names = list(self.core.params)
        self.plist = nn.ParameterList([nn.Parameter(v) for v in self.core.params.values()])
        for n, p in zip(self.names, self.plist):
            self.core.params[n] = p

    def forward(self, x):
        return self.core(x)


def load(path, name):
    neuralop = types.ModuleType("neuralop")
    models = types.ModuleType("neuralop.models")
    models.FNO = models.TFNO = StubFNO
    th = types.ModuleType("torch_harmonics")
    ex = types.ModuleType("torch_harmonics.examples")
    sf = types.ModuleType("torch_harmonics.examples.sfno")
    sf.SphericalFourierNeuralOperatorNet = object
    sys.modules.update({"neuralop": neuralop, "neuralop.models": models, "torch_harmonics": th,
                        "torch_harmonics.examples": ex, "torch_harmonics.examples.sfno": sf})
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def real(t):
    t = t.detach()
    return (torch.view_as_real(t) if t.is_complex() else t).numpy()


def dump(store, tag, stub, **arrays):
    for n, p in zip(stub.names, stub.plist):
        store[f"{tag}/p/{n}"] = real(p)
        store[f"{tag}/g/{n}"] = real(p.grad)
    for k, v in arrays.items():
        store[f"{tag}/{k}"] = v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)


def main():
    ns = load(f"{REF}/nsbench/models/fno/fno.py", "ref_ns_fno")
    dl = load(f"{REF}/dlwpbench/models/fno/fno.py", "ref_dlwp_fno")
    g = torch.Generator().manual_seed(2024)
    out = {}
    # ---- nsbench TFNO2DModule: (T, ctx, tf) incl. T == ctx (one net call), tf == T (no closed loop), ctx 1, tf == ctx - 1 + k
    ns_cases = {"ns_a": dict(B=2, T=8, H=16, W=32, ctx=3, tf=5, hidden=8, n_modes=[8, 8], layers=2),
                "ns_b": dict(B=1, T=4, H=16, W=16, ctx=4, tf=4, hidden=6, n_modes=[4, 6], layers=1),
                "ns_c": dict(B=3, T=6, H=16, W=16, ctx=2, tf=6, hidden=8, n_modes=[6, 6], layers=2),
                "ns_d": dict(B=2, T=7, H=16, W=16, ctx=1, tf=1, hidden=8, n_modes=[6, 6], layers=2),
                "ns_e": dict(B=2, T=9, H=16, W=16, ctx=3, tf=3, hidden=8, n_modes=[8, 8], layers=2)}
    for tag, c in ns_cases.items():
        m = ns.TFNO2DModule(n_modes=c["n_modes"], in_channels=1, hidden_channels=c["hidden"], lifting_channels=16,
                            projection_channels=16, out_channels=1, n_layers=c["layers"], context_size=c["ctx"])
        u = torch.randn(c["B"], c["T"] + 1, 1, c["H"], c["W"], generator=g)
        x, y = u[:, :-1].clone().requires_grad_(True), u[:, 1:].clone()
        yh = m(x, teacher_forcing_steps=c["tf"])
        loss = torch.nn.functional.mse_loss(yh, y)
        loss.backward()
        dump(out, tag, m.fno, x=x, y=y, out=yh, loss=loss, gx=x.grad,
             cfg=np.array([c["ctx"], c["tf"], c["hidden"], c["layers"], c["n_modes"][0], c["n_modes"][1]]))
    # ---- nsbench FNOModule (single frame in, no context)
    m = ns.FNOModule(n_modes=[6, 6], in_channels=1, hidden_channels=8, lifting_channels=16, projection_channels=16,
                     out_channels=1, n_layers=2)
    u = torch.randn(2, 7, 1, 16, 16, generator=g)
    x, y = u[:, :-1].clone(), u[:, 1:].clone()
    yh = m(x, teacher_forcing_steps=3)
    loss = torch.nn.functional.mse_loss(yh, y)
    loss.backward()
    dump(out, "ns_single", m.fno, x=x, y=y, out=yh, loss=loss, cfg=np.array([1, 3, 8, 2, 6, 6]))
    # ---- dlwpbench FNO2DModule
    dl_cases = {"dl_a": dict(B=2, T=5, Cc=2, Cp=1, Cg=3, H=16, W=32, ctx=2), "dl_b": dict(B=2, T=4, Cc=4, Cp=1, Cg=2, H=16, W=16, ctx=1),
                "dl_c": dict(B=1, T=4, Cc=0, Cp=0, Cg=2, H=16, W=16, ctx=1)}
    for tag, c in dl_cases.items():
        m = dl.FNO2DModule(n_modes=[6, 8], constant_channels=c["Cc"], prescribed_channels=c["Cp"], prognostic_channels=c["Cg"],
                           hidden_channels=8, lifting_channels=16, projection_channels=16, n_layers=2, context_size=c["ctx"])
        B, T, H, W, ctx = c["B"], c["T"], c["H"], c["W"], c["ctx"]
        const = torch.randn(B, 1, c["Cc"], H, W, generator=g) if c["Cc"] else None
        presc = torch.randn(B, T, c["Cp"], H, W, generator=g) if c["Cp"] else None
        prog = torch.randn(B, T, c["Cg"], H, W, generator=g)
        target = torch.randn(B, T - ctx, c["Cg"], H, W, generator=g)
        # (i) the published forward, one lead time (T = ctx + 1 never reaches the broken branch)
        one = m(constants=const, prescribed=presc[:, :ctx + 1] if presc is not None else None, prognostic=prog[:, :ctx + 1])
        # (ii) the clean loop over the reference's own _prepare_inputs / self.fno
        outs = []
        for t in range(ctx, T):
            t0 = max(0, t - ctx)
            pt = prog[:, t0:t] if t == ctx else torch.cat([prog[:, t0:ctx], torch.stack(outs, dim=1)[:, -ctx:]], dim=1)
            x_t = m._prepare_inputs(constants=const, prescribed=presc[:, t - ctx:t] if presc is not None else None, prognostic=pt)
            outs.append(pt[:, -1] + m.fno(x_t))
        yh = torch.stack(outs, dim=1)
        assert torch.allclose(one[:, 0], yh[:, 0].detach(), atol=1e-6)
        loss = torch.nn.functional.mse_loss(yh, target)
        loss.backward()
        arrays = dict(prognostic=prog, target=target, out=yh, one_step=one, loss=loss,
                      cfg=np.array([ctx, c["Cc"], c["Cp"], c["Cg"], 8, 2, 6, 8]))
        if const is not None:
            arrays["constants"] = const
        if presc is not None:
            arrays["prescribed"] = presc
        dump(out, tag, m.fno, **arrays)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, f"{os.path.getsize(OUT) / 1e6:.2f} MB,", len(out), "arrays")


if __name__ == "__main__":
    main()
