#!/usr/bin/env python3
"""Per-launch time of SMALL GEMMs measured the way the training steps run them: 50 launches captured in one hipGraph, replayed
(HIP events over 10 replays).  Eager event timing cannot resolve kernels under ~10 us (tools/bench_gemm.py reads 17 - 20 us for
all of them).  Shapes: the SFNO C3 census (profiles/r03_sfno_gemm_census.txt).

    python tools/bench_gemm_graph.py            # environment knobs of csrc/token_ops.hip select the kernel
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402
from dlwp_benchmark_amd.token_ops import _gemm  # noqa: E402

BF = torch.bfloat16
dev = torch.device("cuda:0")
SHAPES = [(8192, 512, 256), (8192, 256, 512), (8192, 256, 256), (32768, 512, 256), (2048, 1536, 384), (8192, 768, 192)]
REPS = 50


def graph_time(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(REPS):
            fn()
    g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / (10 * REPS)


def main():
    L.set_gemm_precision("bf16")
    gen = torch.Generator().manual_seed(0)
    print(f"{'M x N x K':22s} {'y bf16.bf16':>12s} {'y fp32.bf16':>12s} {'gx bf16.bf16':>13s} {'gx fp32.bf16':>13s} {'gW bf16.bf16':>13s}   (us per launch in a graph)")
    for M, N, K in SHAPES:
        x32 = torch.randn(M, K, generator=gen).to(dev)
        w = (torch.randn(N, K, generator=gen) / K ** 0.5).to(dev).to(BF)
        g32 = torch.randn(M, N, generator=gen).to(dev)
        x16, g16 = x32.to(BF), g32.to(BF)
        y = torch.empty(M, N, device=dev, dtype=BF)
        gx = torch.empty(M, K, device=dev, dtype=BF)
        gw = torch.zeros(N, K, device=dev)
        bias = torch.zeros(N, device=dev)
        if os.environ.get("DLWP_BENCH_GEMM_ONLY") == "gW":
            t = graph_time(lambda: _gemm(g16, x16, gw, N, K, M, N, K, K, 1, 0, accumulate=1))
            print(f"{M:6d} x {N:5d} x {K:5d} gW {t:8.2f}")
            continue
        ts = [graph_time(lambda: _gemm(x16, w, y, M, N, K, K, K, N, 0, 1, bias, 1, None, None)),
              graph_time(lambda: _gemm(x32, w, y, M, N, K, K, K, N, 0, 1, bias, 1, None, None)),
              graph_time(lambda: _gemm(g16, w, gx, M, K, N, N, K, K, 0, 0)),
              graph_time(lambda: _gemm(g32, w, gx, M, K, N, N, K, K, 0, 0)),
              graph_time(lambda: _gemm(g16, x16, gw, N, K, M, N, K, K, 1, 0, accumulate=1))]
        print(f"{M:6d} x {N:5d} x {K:5d} " + " ".join(f"{t:12.2f}" for t in ts))


if __name__ == "__main__":
    main()
