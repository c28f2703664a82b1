#!/usr/bin/env python3
"""Token-GEMM micro-benchmark: y = x W^T (the forward product of a Linear layer), gx = g W and gW = g^T x at the shapes of the
C3 / C4 / C5 models, in the three storage / operand modes.  Prints TFLOP/s per launch (HIP events, median of 20).

    python tools/bench_gemm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402
from dlwp_benchmark_amd.token_ops import _gemm  # noqa: E402

BF = torch.bfloat16
SHAPES = [  # (tokens T, out N, in K, label)
    (16200, 3072, 768, "C5 AFNO fc1"), (16200, 768, 3072, "C5 AFNO fc2"),
    (32768, 512, 256, "C3 SFNO mlp fc1 B16"), (32768, 256, 512, "C3 SFNO mlp fc2 B16"),
    (8192, 512, 256, "C3 SFNO mlp fc1 B4"), (8192, 256, 512, "C3 SFNO mlp fc2 B4"),
    (32768, 576, 192, "C4 Pangu qkv (layer1)"), (32768, 768, 192, "C4 Pangu fc1 (layer1, real tokens)"),
    (8192, 1536, 384, "C4 Pangu fc1 (layer2, real tokens)"),
    (8192, 768, 192, "C4 Pangu fc1 (layer1)"), (2048, 1536, 384, "C4 Pangu fc1 (layer2)"),
    (32768, 384, 96, "C4 Swin fc1 (stage 1)"),
]
if os.environ.get("DLWP_BENCH_GEMM_CUBES"):      # structure ceiling of the kernels, away from the model shapes
    SHAPES = [(4096, 4096, 4096, "cube 4k"), (8192, 8192, 8192, "cube 8k")] + SHAPES[:2]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    print(f"{'shape':34s} {'product':8s} " + " ".join(f"{m:>22s}" for m in ("fp32", "bf16 operands", "bf16 storage")))
    for T, N, K, label in SHAPES:
        x = torch.randn(T, K, generator=g).to(dev)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
        gy = torch.randn(T, N, generator=g).to(dev)
        x16, w16, g16 = x.to(BF), w.to(BF), gy.to(BF)
        flops = 2.0 * T * N * K
        rows = {"y=xW^T": [], "gx=gW": [], "gW=g^Tx": []}
        for mode in ("fp32", "bf16", "bf16s"):
            L.set_gemm_precision("fp32" if mode == "fp32" else "bf16")
            xs, ws, gs = (x16, w16, g16) if mode == "bf16s" else (x, w, gy)
            y = torch.empty(T, N, device=dev, dtype=xs.dtype)
            gx = torch.empty(T, K, device=dev, dtype=xs.dtype)
            gw = torch.empty(N, K, device=dev)
            only_gw = os.environ.get("DLWP_BENCH_GEMM_ONLY") == "gW"      # sweeps of the weight-gradient kernel
            rows["y=xW^T"].append(1e9 if only_gw else timeit(lambda: _gemm(xs, ws, y, T, N, K, K, K, N, 0, 1)))
            rows["gx=gW"].append(1e9 if only_gw else timeit(lambda: _gemm(gs, ws, gx, T, K, N, N, K, K, 0, 0)))
            rows["gW=g^Tx"].append(timeit(lambda: _gemm(gs, xs, gw, N, K, T, N, K, K, 1, 0)))
        L.set_gemm_precision("fp32")
        for prod, ts in rows.items():
            print(f"{label + f' {T}x{N}x{K}':34s} {prod:8s} " + " ".join(f"{t:9.1f} us {flops / t / 1e6:7.1f} TF" for t in ts))


if __name__ == "__main__":
    main()
