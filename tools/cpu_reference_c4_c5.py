#!/usr/bin/env python3
"""CPU baseline of BASELINE configs[3] / [4] from the REFERENCE'S OWN CLASSES (build container only: /root/reference does not
exist on the GPU box).  One training step each -- forward rollout (one lead time), MSE, backward, clip_grad_norm_(max_norm = lr),
Adam -- fp32, all host cores, next to one step of the CPU oracle (oracle/*.py) on the same shapes in the same process, so that the
oracle's timing on the GPU box (bench.py's cpu_baseline, kind "port") can be read as the reference's:

  * Pangu-Weather 128 x 256, window (2, 7, 7): src/dlwpbench/models/panguweather/panguweather.py:366-527 (PanguWeather)
  * FourCastNet AFNO 721 x 1440, patch (7, 8), E = 768, depth 12: src/dlwpbench/models/fourcastnet/fourcastnet.py:214-361 (AFNONet)
  * Swin 128 x 256, window 7: the reference SwinTransformer fixes window = stage resolution (swin_transformer.py:542,561) and its
    dlwpbench block cannot pad to a window multiple, so the class cannot be built at BASELINE's window 7.  Timed
    instead: the transformer BODY of that network from the reference's nsbench BasicLayer (src/nsbench/models/swintransformer/
    swin_transformer.py:305-408: 4 blocks at 128 x 256 x 96 + PatchMerging + 4 blocks at 64 x 128 x 192, window 7) -- patch
    embedding, U-decoder and head (about 4 % of the step's flops) are not in the reference number; the oracle is timed on the same
    body AND on the whole step.

    python tools/cpu_reference_c4_c5.py [swin pangu afno721]      -> profiles/r06_cpu_reference_c4_c5.json
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402


def timed_step(params, forward, target, lr, reps=2):          # one warm-up step + one timed step (bench.dlwp_cpu_baseline's protocol)
    opt = torch.optim.Adam(params, lr=lr)
    out = []
    for _ in range(reps):
        t0 = time.perf_counter()
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(forward(), target)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(params, lr)
        opt.step()
        out.append(time.perf_counter() - t0)
    return out


def inputs(w, B):
    g = torch.Generator().manual_seed(1234)
    H, W, Cg, T = w["H"], w["W"], w["Cg"], w["T"]
    return (torch.randn(B, 1, 4, H, W, generator=g), torch.randn(B, T, 1, H, W, generator=g), torch.randn(B, T, Cg, H, W, generator=g),
            torch.randn(B, T - 1, Cg, H, W, generator=g))


def run_pangu():
    import make_pangu_golden
    ref = make_pangu_golden.load_reference()
    w = bench.DLWP_WORKLOADS["pangu"]
    B = w["batch"]
    torch.manual_seed(1234)
    net = ref.PanguWeather(**w["model"])
    net.eval()                      # stochastic depth off (the stub DropPath of the loader refuses training mode); gradients still flow
    c, pr, pg, tg = inputs(w, B)
    ts = timed_step([p for p in net.parameters() if p.requires_grad], lambda: net(constants=c, prescribed=pr, prognostic=pg), tg, w["lr"])
    return {"workload": w["name"], "reference_class": "src/dlwpbench/models/panguweather/panguweather.py:366-527 PanguWeather",
            "batch": B, "reference_s_per_step": round(ts[-1], 3), "reference_samples_per_s": round(B / ts[-1], 4),
            "n_params": sum(p.numel() for p in net.parameters())}


def run_afno721():
    import make_dlwp_afno_golden
    ref = make_dlwp_afno_golden.load_reference()
    w = bench.DLWP_WORKLOADS["afno721"]
    B = w["batch"]
    torch.manual_seed(1234)
    net = ref.AFNONet(**w["model"])
    net.eval()
    c, pr, pg, tg = inputs(w, B)
    ts = timed_step([p for p in net.parameters() if p.requires_grad], lambda: net(constants=c, prescribed=pr, prognostic=pg), tg, 1e-3)
    return {"workload": w["name"], "reference_class": "src/dlwpbench/models/fourcastnet/fourcastnet.py:214-361 AFNONet",
            "batch": B, "reference_s_per_step": round(ts[-1], 3), "reference_samples_per_s": round(B / ts[-1], 4),
            "n_params": sum(p.numel() for p in net.parameters())}


def run_swin():
    import make_swin_golden
    from oracle import swin_ref
    ref = make_swin_golden.load_reference()
    w = bench.DLWP_WORKLOADS["swin"]
    B, E = w["batch"], w["model"]["embed_dim"]
    torch.manual_seed(1234)
    l0 = ref.BasicLayer(dim=E, depth=4, num_heads=4, window_size=7, downsample=ref.PatchMerging)
    l1 = ref.BasicLayer(dim=2 * E, depth=4, num_heads=4, window_size=7)
    for m in (l0, l1):
        m.eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 128 * 256, E, generator=g)
    tg = torch.randn(B, 64 * 128, 2 * E, generator=g)

    def body_ref():
        _, _, _, y, h, ww = l0(x, 128, 256)
        return l1(y, h, ww)[0]
    params = [p for m in (l0, l1) for p in m.parameters() if p.requires_grad]
    t_ref = timed_step(params, body_ref, tg, 1e-3)[-1]
    # the oracle on the same body (same parameters)
    p = {f"layers.0.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in l0.state_dict().items()}
    p.update({f"layers.1.{k}": v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in l1.state_dict().items()})
    leaves = [v for v in p.values() if v.requires_grad]

    def body_oracle():
        _, _, _, y, h, ww = swin_ref.basic_layer(x, p, "layers.0.", 128, 256, 7, 4, 4, True)
        return swin_ref.basic_layer(y, p, "layers.1.", h, ww, 7, 4, 4, False)[0]
    t_or = timed_step(leaves, body_oracle, tg, 1e-3)[-1]
    return {"workload": w["name"], "reference_class": "src/nsbench/models/swintransformer/swin_transformer.py:305-408 BasicLayer x 2 "
            "(+ PatchMerging): the transformer body of the C4 network; the reference SwinTransformer class cannot be built at window 7",
            "batch": B, "reference_body_s_per_step": round(t_ref, 3), "oracle_body_s_per_step": round(t_or, 3),
            "oracle_over_reference_body": round(t_or / t_ref, 3)}


def main():
    which = sys.argv[1:] or ["swin", "pangu", "afno721"]
    threads = os.cpu_count() or 1
    torch.set_num_threads(threads)
    import ctypes
    libc = ctypes.CDLL("libc.so.6")          # as bench.dlwp_cpu_baseline: freed activations stay in the heap
    libc.mallopt(-3, 1 << 30), libc.mallopt(-1, (1 << 31) - 1), libc.mallopt(-2, 1 << 28)
    out = {"host": "build container", "cores": threads, "torch": torch.__version__, "dtype": "fp32",
           "protocol": "training step = forward (one lead time) + MSE + backward + clip_grad_norm_(max_norm = lr) + Adam; one warm-up step, then "
                       "one timed step, glibc malloc kept from returning memory between steps (mallopt); eval() mode = stochastic depth off, "
                       "gradients on", "workloads": []}
    runners = {"swin": run_swin, "pangu": run_pangu, "afno721": run_afno721}
    for wl in which:
        rec = runners[wl]()
        ob = bench.dlwp_cpu_baseline(wl, None, 1.0, clip=bench.DLWP_WORKLOADS[wl].get("lr", 1e-3), threads=threads)
        rec["oracle_s_per_step"] = ob["s_per_step"]
        rec["oracle_samples_per_s"] = ob["value"]
        if "reference_s_per_step" in rec:
            rec["oracle_over_reference"] = round(ob["s_per_step"] / rec["reference_s_per_step"], 3)
        out["workloads"].append(rec)
        print(json.dumps(rec), flush=True)
    path = os.path.join(ROOT, "profiles", "r06_cpu_reference_c4_c5.json")
    if sorted(which) == ["afno721", "pangu", "swin"]:
        json.dump(out, open(path, "w"), indent=1)
        print("wrote", path)


if __name__ == "__main__":
    main()
