#!/usr/bin/env python3
"""Supplementary measurements (not the headline metric): train-step throughput of the AFNO, Swin and Pangu
rollout models built on libdlwpmi's kernels, at the reference's nsbench / dlwpbench shapes.  The step is
forward rollout + MSE + backward (autograd over the HIP ops) + one fused Adam launch on the flat parameter
buffer, captured into a hipGraph by train_engine.GraphedTrainStep (--no-graph: eager dispatch).

    python tools/bench_models.py [fno_dlwp|tfno_dlwp|fno_ctx|afno|afno_tiled|afno_fcn|afno_c5p1|swin|swin_c4|swin_dlwp|sfno|pangu|pangu_c4|all] [--steps N]
"""
import argparse
import json
import sys
import time

import torch

sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from dlwp_benchmark_amd import dlwpbench, nsbench  # noqa: E402
from dlwp_benchmark_amd.train_engine import GraphedTrainStep  # noqa: E402


PRECISION = "fp32"
STORAGE = "fp32"


def run(name, model, make_batch, steps, warmup=3, use_graph=True, call=None, lr=1e-3):
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    inputs, target, B = make_batch(dev)
    step = GraphedTrainStep(model, inputs, target, lr=lr, use_graph=use_graph, call=call)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"model": name, "graph": use_graph, "gemm_precision": PRECISION, "storage": STORAGE, "samples_per_s": round(B * steps / dt, 2),
                      "ms_per_step": round(dt / steps * 1e3, 3), "batch": B, "loss": loss.item(),
                      "n_params": sum(p.numel() for p in model.parameters())}), flush=True)


def run_fno(name, module, step_fn, B, steps, warmup=3):
    """FNO-family modules carry their own fused trainer (fno_engine): time module.train_step + the fused Adam."""
    opt = module.make_optimizer(lr=1e-3)
    for _ in range(warmup):
        step_fn(opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step_fn(opt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"model": name, "graph": True, "gemm_precision": "fp32", "storage": "fp32", "samples_per_s": round(B * steps / dt, 2),
                      "ms_per_step": round(dt / steps * 1e3, 3), "batch": B, "loss": float(loss),
                      "n_params": sum(p.numel() for p in module.parameters())}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="?", default="all")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16"],
                    help="GEMM operand precision (bf16: the reference's autocast arithmetic for C3-C5; fp32 accumulation)")
    ap.add_argument("--storage", default="fp32", choices=["fp32", "bf16"],
                    help="bf16: hidden activations of the token MLPs and a per-step copy of the weights live in HBM as bf16 "
                         "(needs --precision bf16)")
    a = ap.parse_args()
    from dlwp_benchmark_amd import lib as L
    L.set_gemm_precision(a.precision)
    L.set_storage(a.storage)
    global STORAGE
    STORAGE = a.storage
    global PRECISION
    PRECISION = a.precision
    g = torch.Generator().manual_seed(1234)
    dev0 = torch.device("cuda:0")
    if a.which in ("fno_dlwp", "tfno_dlwp"):
        # dlwpbench configs/model/fno.yaml on WeatherBench 5.625 deg shapes (8 prognostic variables), T = 5; tfno_dlwp: the
        # Tucker-factorised weights (rank 0.8) of dlwp TFNO2DModule, fno_dlwp: the dense FNO2DModule
        cls = dlwpbench.TFNO2DModule if a.which == "tfno_dlwp" else dlwpbench.FNO2DModule
        kw = dict(constant_channels=4, prescribed_channels=1, prognostic_channels=8, n_modes=[12, 12], hidden_channels=32,
                  lifting_channels=256, projection_channels=256, n_layers=4, context_size=1)
        if a.which == "tfno_dlwp":
            kw["rank"] = 0.8
        m = cls(**kw).to(dev0)
        B, T = 16, 5
        c = torch.randn(B, 1, 4, 32, 64, generator=g).to(dev0)
        pr = torch.randn(B, T, 1, 32, 64, generator=g).to(dev0)
        pg = torch.randn(B, T, 8, 32, 64, generator=g).to(dev0)
        tg = torch.randn(B, T - 1, 8, 32, 64, generator=g).to(dev0)
        run_fno(f"dlwpbench {cls.__name__} 32x64 hidden32 L4 modes12" + (" rank0.8" if a.which == "tfno_dlwp" else "") + f" B{B} T{T}",
                m, lambda opt: m.train_step(c, pr, pg, tg, optimizer=opt), B, a.steps)
    if a.which in ("fno_ctx",):
        # nsbench configs/model/fno.yaml as shipped: FNOContextModule, n_modes [12, 12, 12] (context 12 = n_modes[0]), 1 layer:
        # a 3-D FNO over (context, H, W) inside the closed-loop rollout, T = 20, teacher forcing 12
        m = nsbench.FNOContextModule(n_modes=[12, 12, 12], in_channels=1, hidden_channels=32, lifting_channels=256,
                                     projection_channels=256, out_channels=1, n_layers=1, context_size=10).to(dev0)
        B, T = 4, 20
        u = torch.randn(B, T + 1, 1, 64, 64, generator=g).to(dev0)
        x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
        run_fno("nsbench FNOContextModule 64x64 modes[12,12,12] hidden32 L1 (3-D FNO over the context) B4 T20",
                m, lambda opt: m.train_step(x, y, 12, optimizer=opt), B, a.steps)
    if a.which in ("afno", "all"):
        # nsbench configs/model/fourcastnet.yaml with the paper runs' context 10 (train_commands.txt:112-115), T=20, tf=10
        m = nsbench.AFNONet(img_height=64, img_width=64, patch_size=(4, 4), in_chans=1, out_chans=1, embed_dim=64, depth=4,
                            mlp_ratio=4.0, num_blocks=4, context_size=10)

        def batch(dev):
            u = torch.randn(4, 21, 1, 64, 64, generator=g).to(dev)
            return {"x": u[:, :-1].contiguous()}, u[:, 1:].contiguous(), 4
        run("nsbench AFNONet 64x64 p4 E64 depth4 ctx10 T20", m, batch, a.steps, use_graph=not a.no_graph,
            call=lambda mod, kw: mod(kw["x"], 10))
    if a.which in ("afno_tiled",):
        # same nsbench config through the general-grid (batched GEMM) AFNO2D path
        m = nsbench.AFNONet(img_height=64, img_width=64, patch_size=(4, 4), in_chans=1, out_chans=1, embed_dim=64, depth=4,
                            mlp_ratio=4.0, num_blocks=4, context_size=10)
        for mod in m.modules():
            if hasattr(mod, "path"):
                mod.path = "tiled"

        def batch(dev):
            u = torch.randn(4, 21, 1, 64, 64, generator=g).to(dev)
            return {"x": u[:, :-1].contiguous()}, u[:, 1:].contiguous(), 4
        run("nsbench AFNONet 64x64 p4 E64 depth4 ctx10 T20 (tiled AFNO2D)", m, batch, a.steps, use_graph=not a.no_graph,
            call=lambda mod, kw: mod(kw["x"], 10))
    if a.which in ("afno_fcn",):
        # FourCastNet-paper scale (BASELINE C5 grid): dlwpbench AFNONet 720x1440, patch 8, E=768, depth 12, 16 blocks
        m = dlwpbench.AFNONet(img_height=720, img_width=1440, patch_size=(8, 8), constant_channels=4, prescribed_channels=1,
                              prognostic_channels=8, embed_dim=768, depth=12, mlp_ratio=4.0, num_blocks=16, context_size=1)

        def batch(dev):
            kw = dict(constants=torch.randn(1, 1, 4, 720, 1440, generator=g).to(dev),
                      prescribed=torch.randn(1, 2, 1, 720, 1440, generator=g).to(dev),
                      prognostic=torch.randn(1, 2, 8, 720, 1440, generator=g).to(dev))
            return kw, torch.randn(1, 1, 8, 720, 1440, generator=g).to(dev), 1
        run("dlwpbench AFNONet 720x1440 p8 E768 depth12 nb16 B1 T2", m, batch, a.steps, use_graph=not a.no_graph)
    if a.which in ("afno_c5p1",):
        # BASELINE C5 as worded: the shipped dlwpbench fourcastnet.yaml (patch [1, 1], E=64, depth 4, 4 blocks) on the ERA5 0.25 deg
        # grid 721 x 1440 = 1.04 M tokens: the AFNO filter runs on the LDS-staged rFFT2 kernels (721 = 7 * 103: generic prime pass)
        m = dlwpbench.AFNONet(img_height=721, img_width=1440, patch_size=(1, 1), constant_channels=4, prescribed_channels=1,
                              prognostic_channels=8, embed_dim=64, depth=4, mlp_ratio=4.0, num_blocks=4, context_size=1)

        def batch(dev):
            kw = dict(constants=torch.randn(1, 1, 4, 721, 1440, generator=g).to(dev),
                      prescribed=torch.randn(1, 2, 1, 721, 1440, generator=g).to(dev),
                      prognostic=torch.randn(1, 2, 8, 721, 1440, generator=g).to(dev))
            return kw, torch.randn(1, 1, 8, 721, 1440, generator=g).to(dev), 1
        run("dlwpbench AFNONet 721x1440 p1 E64 depth4 nb4 (shipped fourcastnet.yaml on the C5 grid) B1 T2", m, batch, a.steps,
            use_graph=not a.no_graph)
    if a.which in ("swin", "all"):
        m = nsbench.SwinTransformer(context_size=10, pretrain_img_size=64, patch_size=2, in_chans=1, out_chans=1,
                                    embed_dim=40, depths=[4, 4], num_heads=[4, 4], drop_path_rate=0.0)

        def batch(dev):
            u = torch.randn(4, 21, 1, 64, 64, generator=g).to(dev)
            return {"x": u[:, :-1].contiguous()}, u[:, 1:].contiguous(), 4
        run("nsbench SwinTransformer 64x64 p2 E40 depths[4,4] ctx10 T20", m, batch, a.steps, use_graph=not a.no_graph,
            call=lambda mod, kw: mod(kw["x"], 10))
    if a.which in ("swin_c4",):
        # BASELINE C4 shapes: dlwpbench SwinTransformer 128x256, window 7 (extra kwarg), E=96, depths [4,4], heads [4,4]
        m = dlwpbench.SwinTransformer(constant_channels=4, prescribed_channels=1, prognostic_channels=8, context_size=1,
                                      img_height=128, img_width=256, patch_size=1, embed_dim=96, depths=[4, 4],
                                      num_heads=[4, 4], drop_path_rate=0.2, window_size=7)      # (bench.py's configuration)

        def batch(dev):
            kw = dict(constants=torch.randn(2, 1, 4, 128, 256, generator=g).to(dev),
                      prescribed=torch.randn(2, 2, 1, 128, 256, generator=g).to(dev),
                      prognostic=torch.randn(2, 2, 8, 128, 256, generator=g).to(dev))
            return kw, torch.randn(2, 1, 8, 128, 256, generator=g).to(dev), 2
        run("dlwpbench SwinTransformer 128x256 window7 E96 depths[4,4] (C4) B2 T2", m, batch, a.steps, use_graph=not a.no_graph)
    if a.which in ("swin_dlwp",):
        # dlwpbench configs/model/swintransformer.yaml as shipped: 32x64, patch 1, whole-map windows (N = 2048 at stage 0)
        m = dlwpbench.SwinTransformer(constant_channels=4, prescribed_channels=1, prognostic_channels=8, context_size=1,
                                      img_height=32, img_width=64, patch_size=1, embed_dim=96, depths=[4, 4], num_heads=[4, 4],
                                      drop_path_rate=0.0)

        def batch(dev):
            kw = dict(constants=torch.randn(4, 1, 4, 32, 64, generator=g).to(dev),
                      prescribed=torch.randn(4, 5, 1, 32, 64, generator=g).to(dev),
                      prognostic=torch.randn(4, 5, 8, 32, 64, generator=g).to(dev))
            return kw, torch.randn(4, 4, 8, 32, 64, generator=g).to(dev), 4
        run("dlwpbench SwinTransformer 32x64 p1 E96 depths[4,4] whole-map windows B4 T5", m, batch, a.steps,
            use_graph=not a.no_graph)
    if a.which in ("pangu_c4",):
        m = dlwpbench.PanguWeather(constant_channels=4, prescribed_channels=1, prognostic_channels=8, embed_dim=192,
                                   num_heads=(6, 12, 12, 6), window_size=(2, 7, 7), patch_size=(1, 1), n_lat=128, n_lon=256,
                                   context_size=1)

        def batch(dev):
            kw = dict(constants=torch.randn(1, 1, 4, 128, 256, generator=g).to(dev),
                      prescribed=torch.randn(1, 2, 1, 128, 256, generator=g).to(dev),
                      prognostic=torch.randn(1, 2, 8, 128, 256, generator=g).to(dev))
            return kw, torch.randn(1, 1, 8, 128, 256, generator=g).to(dev), 1
        run("dlwpbench PanguWeather 128x256 window(2,7,7) E192 (C4) B1 T2", m, batch, a.steps, use_graph=not a.no_graph, lr=1e-4)
    if a.which in ("sfno", "all"):
        # BASELINE configs[2] (C3): dlwpbench SFNO2DModule, configs/model/sfno.yaml with 5 prognostic variables, 32x64
        for B in ((4, 16) if a.which == "sfno" else (16,)):
            m = dlwpbench.SFNO2DModule(constant_channels=4, prescribed_channels=1, prognostic_channels=5, grid="equiangular",
                                       num_layers=4, scale_factor=1, embed_dim=256, context_size=1, height=32, width=64,
                                       big_skip=True, pos_embed=True, use_mlp=True, normalization_layer="none")

            def batch(dev, B=B):
                kw = dict(constants=torch.randn(B, 1, 4, 32, 64, generator=g).to(dev),
                          prescribed=torch.randn(B, 5, 1, 32, 64, generator=g).to(dev),
                          prognostic=torch.randn(B, 5, 5, 32, 64, generator=g).to(dev))
                return kw, torch.randn(B, 4, 5, 32, 64, generator=g).to(dev), B
            run(f"dlwpbench SFNO2DModule 32x64 E256 L4 (C3) B{B} T5", m, batch, a.steps, use_graph=not a.no_graph)
    if a.which in ("pangu", "all"):
        # dlwpbench configs/model/panguweather.yaml at 32x64 with 5 prognostic variables (BASELINE configs[2] shapes)
        m = dlwpbench.PanguWeather(constant_channels=4, prescribed_channels=1, prognostic_channels=5, embed_dim=192,
                                   num_heads=(6, 12, 12, 6), window_size=(2, 6, 12), patch_size=(1, 1), n_lat=32, n_lon=64,
                                   context_size=1)
        m.eval_drop_path = True

        def batch(dev):
            kw = dict(constants=torch.randn(1, 1, 4, 32, 64, generator=g).to(dev),
                      prescribed=torch.randn(1, 5, 1, 32, 64, generator=g).to(dev),
                      prognostic=torch.randn(1, 5, 5, 32, 64, generator=g).to(dev))
            return kw, torch.randn(1, 4, 5, 32, 64, generator=g).to(dev), 1
        run("dlwpbench PanguWeather 32x64 E192 window(2,6,12) B1 T5", m, batch, a.steps, use_graph=not a.no_graph, lr=1e-4)


if __name__ == "__main__":
    main()
