#!/usr/bin/env python3
"""What the vendor GEMM library (hipBLASLt / rocBLAS behind torch.matmul) reaches on the C4 / C5 product shapes: a yardstick for
the hand-written kernels of csrc/token_ops.hip, not a code path of the library (nothing under dlwp_benchmark_amd/ calls torch.matmul)."""
import torch

dev = torch.device("cuda:0")
shapes = [("C5 fc1  y = x W^T", 16200, 3072, 768, "nt"), ("C5 fc2  y = h W^T", 16200, 768, 3072, "nt"),
          ("C5 gh = g W2", 16200, 3072, 768, "nn"), ("C5 gx = gh W1", 16200, 768, 3072, "nn"),
          ("C5 gW = g^T x", 3072, 768, 16200, "tn"), ("Pangu fc1 L2", 8192, 1536, 384, "nt"), ("Pangu fc1 L1", 32768, 768, 192, "nt"),
          ("Swin fc1", 65536, 384, 96, "nt")]
for name, M, N, K, lay in shapes:
    a = torch.randn((K, M) if lay[0] == "t" else (M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((N, K) if lay[1] == "t" else (K, N), device=dev, dtype=torch.bfloat16)
    A = a.t() if lay[0] == "t" else a
    Bm = b.t() if lay[1] == "t" else b
    for _ in range(5):
        c = A @ Bm
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        c = A @ Bm
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:18s} M={M:6d} N={N:5d} K={K:6d} {lay}: {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TFLOP/s (plain product, bf16 out, no epilogue)")
