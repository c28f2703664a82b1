#!/usr/bin/env python3
"""Race screen of the two-group 256 x 256 GEMM kernels (csrc/token_ops.hip, gemm_p8_kernel / gemm_p8_tn_kernel): the same product
launched many times at several shapes, every result compared bit for bit with the first and against a float64 product.  A read
that slips ahead of its LDS-DMA (or a refill ahead of the other group's read) would show as a mismatch that comes and goes.

    python tools/race_screen_gemm.py [launches per shape = 40]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402
from dlwp_benchmark_amd.token_ops import _gemm  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
BF = torch.bfloat16


def main():
    L.set_gemm_precision("bf16")
    L.set_gemm_tile256(1)
    bad = 0
    for (M, N, K) in [(4096, 4096, 4096), (16200, 768, 3072), (16200, 3072, 768), (1000, 520, 192), (2048, 2048, 8192), (300, 264, 64)]:
        g = torch.Generator().manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g).to(dev).to(BF)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(BF)
        bias = torch.randn(N, generator=g).to(dev)
        y0 = None
        for i in range(REPS):
            y = torch.empty(M, N, device=dev)
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 0, None, None)
            if y0 is None:
                y0 = y.clone()
                ref = (x[:256].double() @ w.double().T + bias.double())
                err = ((y0[:256].double() - ref).abs().max() / ref.abs().max()).item()
            elif not torch.equal(y, y0):
                bad += 1
        print(f"y = x W^T   {M} x {N} x {K}: {REPS} launches, mismatching launches {bad}, rel err of rows 0-255 vs float64 {err:.2e}")
    for (M, N, K) in [(3072, 768, 16200), (768, 3072, 16200), (512, 512, 8192), (264, 520, 3000)]:
        g = torch.Generator().manual_seed(M + K)
        gy = torch.randn(K, M, generator=g).to(dev).to(BF)
        x = torch.randn(K, N, generator=g).to(dev).to(BF)
        w0 = None
        for i in range(REPS):
            gw = torch.zeros(M, N, device=dev)
            _gemm(gy, x, gw, M, N, K, M, N, N, 1, 0, accumulate=1)
            if w0 is None:
                w0 = gw.clone()
                ref = gy[:, :128].double().T @ x.double()
                err = ((w0[:128].double() - ref).abs().max() / ref.abs().max()).item()
            elif not torch.equal(gw, w0):
                bad += 1
        print(f"gW = g^T x  {M} x {N} x {K}: {REPS} launches, mismatching launches {bad}, rel err of rows 0-127 vs float64 {err:.2e}")
    print("race screen:", "CLEAN" if bad == 0 else f"{bad} MISMATCHES")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
