#!/usr/bin/env python3
"""The four activation products of one FourCastNet (C5) MLP WITH the epilogues the training step uses, bf16 operands + bf16
storage (the calls of token_ops._MlpFn, reference Mlp.forward: src/nsbench/models/fourcastnet/fourcastnet.py:40-56):

    fc1   h  = gelu(x W1^T + b1), z = x W1^T + b1 stored too     16200 x 3072 x 768     bias + GELU + two bf16 outputs
    fc2   y  = h W2^T + b2 + residual                            16200 x 768 x 3072     bias + fp32 residual, fp32 output
    gh    gh = (g W2) * gelu'(z)                                 16200 x 3072 x 768     bf16 z read, GELU', bf16 output
    gx    gx = gh W1                                             16200 x 768 x 3072     fp32 output

Prints microseconds and TFLOP/s per launch (HIP events on the current stream, median of 20) -- the numbers DESIGN.md quotes for
C5's GEMMs come from here, not from the epilogue-free tools/bench_gemm.py.

    python tools/bench_gemm_epilogue.py [T]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402
from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched  # noqa: E402

BF = torch.bfloat16


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
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 16200
    E, Hd = 768, 3072
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(T, E, generator=g).to(dev).to(BF)
    w1 = (torch.randn(Hd, E, generator=g) / E ** 0.5).to(dev).to(BF)
    w2 = (torch.randn(E, Hd, generator=g) / Hd ** 0.5).to(dev).to(BF)
    b1, b2 = torch.randn(Hd, generator=g).to(dev), torch.randn(E, generator=g).to(dev)
    res = torch.randn(T, E, generator=g).to(dev)
    gy = torch.randn(T, E, generator=g).to(dev).to(BF)
    z, h, gh = (torch.empty(T, Hd, device=dev, dtype=BF) for _ in range(3))
    y, gx = torch.empty(T, E, device=dev), torch.empty(T, E, device=dev)
    L.set_gemm_precision("bf16")
    flops = 2.0 * T * E * Hd
    calls = [
        ("fc1  bias + GELU + z, h (bf16)", lambda: _gemm(x, w1, h, T, Hd, E, E, E, Hd, 0, 1, b1, 1, z, None)),
        ("fc1  plain (bf16 out)", lambda: _gemm(x, w1, h, T, Hd, E, E, E, Hd, 0, 1)),
        ("fc2  bias + residual (fp32 out)", lambda: _gemm(h, w2, y, T, E, Hd, Hd, Hd, E, 0, 1, b2, 0, None, res)),
        ("gh   (g W2) * GELU'(z) (bf16 out)", lambda: _gemm_batched(gy, w2, gh, T, Hd, E, E, Hd, Hd, 0, 0, act=4, residual=z)),
        ("gx   gh W1 (fp32 out)", lambda: _gemm(gh, w1, gx, T, E, Hd, Hd, E, E, 0, 0)),
    ]
    total = 0.0
    for label, fn in calls:
        us = timeit(fn)
        if "plain" not in label:
            total += us
        print(f"T={T} {label:36s} {us:8.1f} us  {flops / us * 1e-6:7.1f} TFLOP/s")
    print(f"T={T} the four activation products of one MLP: {total:.1f} us ({4 * flops / total * 1e-6:.1f} TFLOP/s)")


if __name__ == "__main__":
    main()
