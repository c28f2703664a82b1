#!/usr/bin/env python3
"""Does the bf16 path TRAIN like the fp32 path?  (round-5 verdict, item 3: the benchmarked bf16-storage paths were only ever compared
with themselves at loose bars, and the AFNO spectrum window is stored as bf16 between the two transforms although the reference
computes the spectral mixer in fp32: src/nsbench/models/fourcastnet/fourcastnet.py:80-81,120-124.)

A learnable synthetic forecasting task with known dynamics, so that closed-loop RMSE means something: band-limited random
fields advected in longitude by a fixed (fractional) shift per step, rotated in channel space, damped by a "constant" field and
forced by the "prescribed" field,

    x_{t+1} = (1 - 0.05 c_0) * R_theta shift_lon(x_t, s_c) + 0.1 m_c f_t,

fresh samples every step (no over-fitting), Adam, no gradient clipping (as the paper runs, nsbench train_commands.txt:83).  Per
family and seed the SAME data stream and the SAME initial weights are trained in every arithmetic mode:

    fp32        the HIP fp32 path (exact-fp32 MFMA; oracle-equal at 1e-4, tests/test_gpu_sfno.py, test_gpu_afno.py)
    bf16        bf16 operands + bf16 storage: what bench.py measures for configs[2] - [4]
    bf16_fp32spectra   (AFNO only) the same with the spectrum window kept fp32 between the transforms (the default since round 6)
    bf16_spectra       (AFNO only) bf16 storage INCLUDING the spectrum window (round 5's default, DLWP_AFNO_SPECTRA_BF16=1)
    bf16_operands      (SFNO only) bf16 operands with fp32 storage: what the operand rounding alone costs

and then rolled out closed loop on held-out fields (lead times 1 .. 4 from one observed frame) at several points of the training
(the bf16 rounding floor only shows once the error is small).  Families: SFNO2DModule at the sfno.yaml widths on 32 x 64 (BASELINE
configs[2]) and AFNONet with the benchmarked FourCastNet widths (E = 768, 16 blocks, patch 8; depth 6 of 12) on a 360 x 720 grid
(45 x 90 tokens: the rFFT2 path of configs[4], a quarter of its tokens).

    python tools/bf16_training_quality.py [--steps-sfno 2000 --steps-afno 1500 --seeds 3]   -> profiles/r06_bf16_training_quality.json
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from dlwp_benchmark_amd import afno_tiled, dlwpbench, lib as L  # noqa: E402
from dlwp_benchmark_amd.train_engine import GraphedTrainStep, refresh_bf16_weights  # noqa: E402


class AdvectionTask:
    """The synthetic dynamics above on an H x W lat-lon grid with Cg prognostic channels, 4 constants and 1 prescribed channel."""

    def __init__(self, H, W, Cg, device, kmax=6, shift=1.5, theta=0.3):
        self.H, self.W, self.Cg, self.dev, self.kmax, self.theta = H, W, Cg, device, kmax, theta
        lat = torch.linspace(-math.pi / 2, math.pi / 2, H, device=device)[:, None]
        lon = torch.linspace(0, 2 * math.pi, W + 1, device=device)[None, :-1]
        self.lat, self.lon = lat, lon
        self.const = torch.stack([0.5 + 0.5 * torch.cos(lat) * torch.sin(2 * lon), torch.sin(lat).expand(H, W), torch.cos(lon).expand(H, W),
                                  (torch.cos(lat) * torch.cos(3 * lon))])                                         # [4, H, W], c_0 in [0, 1]
        k = torch.arange(W // 2 + 1, device=device)
        s = shift * (1.0 + 0.25 * torch.arange(Cg, device=device))                                            # grid columns per step
        self.phase = torch.exp(-2j * math.pi * k[None, :] * s[:, None] / W)                                      # [Cg, W/2+1]
        R = torch.eye(Cg, device=device)
        c, sn = math.cos(theta), math.sin(theta)
        for a in range(0, Cg - 1, 2):
            R[a, a], R[a, a + 1], R[a + 1, a], R[a + 1, a + 1] = c, -sn, sn, c
        self.R = R
        self.m = torch.linspace(1.0, -1.0, Cg, device=device)[:, None, None]
        self.taper = torch.cos(lat).clamp_min(0.0) ** 0.5

    def initial(self, B, gen):
        H, W, K = self.H, self.W, self.kmax
        spec = torch.zeros(B, self.Cg, H, W // 2 + 1, dtype=torch.complex64, device=self.dev)
        ky = torch.cat([torch.arange(0, K + 1), torch.arange(-K, 0)]).to(self.dev)
        amp = 1.0 / (1.0 + (ky[:, None].float() ** 2 + torch.arange(K + 1, device=self.dev)[None, :].float() ** 2)) ** 0.75
        re = torch.randn(B, self.Cg, 2 * K + 1, K + 1, generator=gen, device=self.dev)
        im = torch.randn(B, self.Cg, 2 * K + 1, K + 1, generator=gen, device=self.dev)
        spec[:, :, ky % H, :K + 1] = torch.complex(re, im) * amp
        x = torch.fft.irfft2(spec, s=(H, W)) * self.taper
        return x / x.std(dim=(1, 2, 3), keepdim=True)

    def forcing(self, B, T, gen):
        ph = 2 * math.pi * torch.rand(B, 1, 1, 1, generator=gen, device=self.dev)
        t = torch.arange(T, device=self.dev)[None, :, None, None]
        return (torch.cos(self.lat) * torch.sin(2 * (self.lon - 0.2 * t) + ph)).unsqueeze(2)                    # [B, T, 1, H, W]

    def step(self, x, f):
        xs = torch.fft.irfft(torch.fft.rfft(x, dim=-1) * self.phase[None, :, None, :], n=self.W, dim=-1)
        xr = torch.einsum("ij,bjhw->bihw", self.R, xs)
        return (1.0 - 0.05 * self.const[0]) * xr + 0.1 * self.m * f

    def sample(self, B, T, gen):
        """(constants [B,1,4,H,W], prescribed [B,T,1,H,W], prognostic [B,T,Cg,H,W]); frame t+1 follows from frame t and prescribed t."""
        f = self.forcing(B, T, gen)
        frames = [self.initial(B, gen)]
        for t in range(T - 1):
            frames.append(self.step(frames[-1], f[:, t]))
        return self.const[None, None].expand(B, 1, 4, self.H, self.W).contiguous(), f.contiguous(), torch.stack(frames, dim=1).contiguous()


FAMILIES = {
    "sfno": dict(cls="SFNO2DModule", H=32, W=64, Cg=5, B=16, T=5, lr=1e-3, shift=0.75,
                 model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=5, grid="equiangular", num_layers=4, scale_factor=1,
                            embed_dim=256, context_size=1, height=32, width=64, big_skip=True, pos_embed=True, use_mlp=True,
                            normalization_layer="none"),
                 modes=("fp32", "bf16", "bf16_operands")),
    # lr 2e-4: at 5e-4 (first run of this script, kept in the JSON as "first_run") the 72 M parameter network's loss jumped by 4 x between
    # checkpoints and its closed-loop error sat ABOVE persistence in every arithmetic: nothing to compare
    "afno": dict(cls="AFNONet", H=360, W=720, Cg=8, B=4, T=2, lr=2e-4, shift=3.0,
                 model=dict(img_height=360, img_width=720, patch_size=(8, 8), constant_channels=4, prescribed_channels=1,
                            prognostic_channels=8, embed_dim=768, depth=6, mlp_ratio=4.0, num_blocks=16, context_size=1),
                 modes=("fp32", "bf16_fp32spectra", "bf16_spectra")),
    # round 6, second half: the C4 families with the bf16 tensor paths that were added then (window-layout bf16 attention tensors, the
    # LayerNorm backward's scaled bf16 second output under stochastic depth, transposed weight copies): the benchmarked widths on a
    # quarter-size grid whose stage maps are multiples of the window
    "swin": dict(cls="SwinTransformer", H=56, W=112, Cg=8, B=4, T=2, lr=5e-4, shift=1.5,
                 model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=8, context_size=1, img_height=56, img_width=112,
                            patch_size=1, embed_dim=96, depths=[2, 2], num_heads=[4, 4], drop_path_rate=0.1, window_size=7),
                 modes=("fp32", "bf16")),
    "pangu": dict(cls="PanguWeather", H=56, W=112, Cg=8, B=2, T=2, lr=3e-4, shift=1.5,
                  model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=8, embed_dim=96, num_heads=(3, 6, 6, 3),
                             window_size=(2, 7, 7), patch_size=(1, 1), n_lat=56, n_lon=112, context_size=1),
                  modes=("fp32", "bf16")),
}


def set_mode(mode):
    L.set_storage("fp32")
    L.set_gemm_precision("fp32" if mode == "fp32" else "bf16")
    if mode not in ("fp32", "bf16_operands"):
        L.set_storage("bf16")
    afno_tiled._SPECTRA_BF16 = mode == "bf16_spectra"


def evaluate(model, task, batches, B, T_eval, seed):
    """closed-loop RMSE per lead time over held-out samples, in the arithmetic mode that is set (bf16 storage: through the engine's
    bf16 weight copies, as inside a train step), and the persistence forecast's RMSE for scale."""
    gen = torch.Generator(device=task.dev).manual_seed(10_000 + seed)
    se = torch.zeros(T_eval - 1, device=task.dev, dtype=torch.float64)
    se_p = torch.zeros_like(se)
    n = 0
    refresh_bf16_weights(model)
    prev, L.SHADOW_ACTIVE = L.SHADOW_ACTIVE, True
    try:
        for _ in range(batches):
            c, f, x = task.sample(B, T_eval, gen)
            with torch.no_grad():
                out = model(constants=c, prescribed=f, prognostic=x[:, :1].expand(-1, T_eval, -1, -1, -1).contiguous())
            truth = x[:, 1:]
            se += ((out.double() - truth.double()) ** 2).mean(dim=(0, 2, 3, 4))
            se_p += ((x[:, :1].double() - truth.double()) ** 2).mean(dim=(0, 2, 3, 4))
            n += 1
    finally:
        L.SHADOW_ACTIVE = prev
    return (se / n).sqrt().tolist(), (se_p / n).sqrt().tolist()


def run(family, mode, seed, steps, device, log_every=0, eval_at=()):
    cfg = FAMILIES[family]
    set_mode(mode)
    task = AdvectionTask(cfg["H"], cfg["W"], cfg["Cg"], device, shift=cfg["shift"])
    torch.manual_seed(1000 + seed)
    model = getattr(dlwpbench, cfg["cls"])(**cfg["model"]).to(device).train()
    gen = torch.Generator(device=device).manual_seed(seed)
    c, f, x = task.sample(cfg["B"], cfg["T"], gen)
    step = GraphedTrainStep(model, dict(constants=c, prescribed=f, prognostic=x), x[:, 1:].contiguous(), lr=cfg["lr"], clip_max_norm=None)
    losses, at = [], {}
    t0 = time.perf_counter()
    for i in range(steps):
        c, f, x = task.sample(cfg["B"], cfg["T"], gen)
        loss = step(dict(constants=c, prescribed=f, prognostic=x), x[:, 1:].contiguous())
        if i % 50 == 49 or i == steps - 1:
            losses.append(round(loss.item(), 6))
            if log_every and i % log_every == log_every - 1:
                print(f"  {family} {mode} seed {seed} step {i + 1}: loss {losses[-1]:.5f}", flush=True)
        if (i + 1) in eval_at and i + 1 < steps:
            r_, _ = evaluate(model, task, batches=4, B=cfg["B"], T_eval=5, seed=seed)
            at[str(i + 1)] = {"closed_loop_rmse": (sum(v * v for v in r_) / len(r_)) ** 0.5, "lead1_rmse": r_[0]}
            model.train()
    torch.cuda.synchronize()
    train_s = time.perf_counter() - t0
    rmse, pers = evaluate(model, task, batches=4, B=cfg["B"], T_eval=5, seed=seed)
    at[str(steps)] = {"closed_loop_rmse": (sum(v * v for v in rmse) / len(rmse)) ** 0.5, "lead1_rmse": rmse[0]}
    del step, model
    torch.cuda.empty_cache()
    return {"family": family, "mode": mode, "seed": seed, "steps": steps, "train_s": round(train_s, 2), "loss_every_50": losses,
            "closed_loop_rmse_per_lead": [round(v, 6) for v in rmse], "closed_loop_rmse": round(sum(v * v for v in rmse) / len(rmse), 8) ** 0.5,
            "lead1_rmse": rmse[0], "at_step": at,
            "persistence_rmse": round(sum(v * v for v in pers) / len(pers), 8) ** 0.5, "persistence_lead1_rmse": pers[0]}


def summarise(runs):
    """per family and training length: mean over seeds of each mode's closed-loop / lead-1 RMSE, its ratio to the fp32 mean and the
    seed-paired ratios (same data stream, same initial weights)."""
    out = {}
    for fam in sorted({r["family"] for r in runs}):
        fr = [r for r in runs if r["family"] == fam]
        modes = sorted({r["mode"] for r in fr}, key=lambda m: (m != "fp32", m))
        fam_out = {"persistence_rmse": fr[0]["persistence_rmse"], "persistence_lead1_rmse": fr[0]["persistence_lead1_rmse"]}
        for step in sorted({k for r in fr for k in r["at_step"]}, key=int):
            for metric in ("closed_loop_rmse", "lead1_rmse"):
                ref = {r["seed"]: r["at_step"][step][metric] for r in fr if r["mode"] == "fp32" and step in r["at_step"]}
                row = {}
                for m in modes:
                    vals = {r["seed"]: r["at_step"][step][metric] for r in fr if r["mode"] == m and step in r["at_step"]}
                    if not vals:
                        continue
                    v = list(vals.values())
                    mean = sum(v) / len(v)
                    sd = (sum((a - mean) ** 2 for a in v) / max(1, len(v) - 1)) ** 0.5
                    row[m] = {"mean": mean, "sd_over_seeds": sd, "by_seed": vals}
                    if ref:
                        row[m]["over_fp32_mean"] = mean / (sum(ref.values()) / len(ref))
                        row[m]["paired_ratio_by_seed"] = {s_: vals[s_] / ref[s_] for s_ in vals if s_ in ref}
                fam_out[f"after_{step}_steps.{metric}"] = row
        out[fam] = fam_out
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps-sfno", type=int, default=2000)
    ap.add_argument("--steps-afno", type=int, default=1500)
    ap.add_argument("--steps-c4", type=int, default=800, help="training steps of the swin / pangu families")
    ap.add_argument("--seeds", type=int, default=3)
    ap.add_argument("--families", default="sfno,afno")
    ap.add_argument("--modes", default=None, help="comma list: only these modes (default: all of the family)")
    ap.add_argument("--merge", default=None, help="a JSON written by an earlier call whose runs are kept (same protocol, other modes)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r06_bf16_training_quality.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    runs = []
    if a.merge and os.path.isfile(a.merge):
        runs = [r for r in json.load(open(a.merge))["runs"] if "at_step" in r]
    for fam in a.families.split(","):
        steps = a.steps_sfno if fam == "sfno" else a.steps_afno if fam == "afno" else a.steps_c4
        eval_at = tuple(steps // d for d in (8, 4, 2))
        for seed in range(a.seeds):
            for mode in FAMILIES[fam]["modes"]:
                if a.modes and mode not in a.modes.split(","):
                    continue
                if any(r["family"] == fam and r["mode"] == mode and r["seed"] == seed and r["steps"] == steps for r in runs):
                    continue
                r = run(fam, mode, seed, steps, dev, log_every=500, eval_at=eval_at)
                runs.append(r)
                print(json.dumps({k: r[k] for k in ("family", "mode", "seed", "train_s", "closed_loop_rmse", "lead1_rmse", "persistence_rmse")}),
                      flush=True)
                json.dump({"runs": runs}, open(a.out + ".partial", "w"))
    set_mode("fp32")
    doc = {"task": __doc__.split("\n\n")[1], "protocol": {"steps": {"sfno": a.steps_sfno, "afno": a.steps_afno, "swin": a.steps_c4, "pangu": a.steps_c4}, "seeds": a.seeds,
                                                           "families": {k: {kk: vv for kk, vv in v.items() if kk != "modes"} for k, v in FAMILIES.items()},
                                                           "evaluation": "closed loop from ONE observed frame, lead times 1 - 4, 4 held-out batches, RMSE over "
                                                                         "all lead times; each mode evaluated in its own arithmetic"},
           "summary": summarise(runs), "runs": runs}
    json.dump(doc, open(a.out, "w"), indent=1)
    if os.path.isfile(a.out + ".partial"):
        os.remove(a.out + ".partial")
    print(json.dumps(doc["summary"], indent=1))


if __name__ == "__main__":
    main()
