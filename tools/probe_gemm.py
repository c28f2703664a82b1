#!/usr/bin/env python3
"""GEMM throughput vs shape and operand precision (y = gelu(x W^T + b), fp32 tensors in HBM)."""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L
from dlwp_benchmark_amd.token_ops import _gemm
dev = torch.device("cuda:0")
for prec in ("fp32", "bf16"):
    L.set_gemm_precision(prec)
    for (M, N, K) in [(2048, 512, 256), (8192, 512, 256), (32768, 512, 256), (131072, 512, 256), (8192, 256, 512), (8192, 512, 512),
                      (16384, 3072, 768), (16384, 768, 3072), (5184, 576, 192), (4096, 4096, 4096)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev); z = torch.empty(M, N, device=dev)
        f = lambda: _gemm(x, w, y, M, N, K, K, K, N, 0, 1, b, 1, z, None)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        gb = (M * K + N * K + 2 * M * N) * 4 / 1e9
        print(f"{prec} {M:7d} x {N:5d} x {K:5d}: {us:9.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s  {gb / us * 1e6 / 1e3:6.2f} TB/s io")
