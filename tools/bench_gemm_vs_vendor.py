#!/usr/bin/env python3
"""Plain bf16 products of the C4 / C5 shapes: every kernel family of csrc/token_ops.hip that can take the product (forced through the
tuning registry) next to the vendor library behind torch.matmul (hipBLASLt / rocBLAS: a yardstick, not a code path of the product),
all timed the same way -- `reps` back-to-back launches between two events, `rounds` interleaved rounds per variant in ONE process,
median and minimum reported (cdna_hip_programming.md section 5.4 rule 24).  Random operands (rule 25).

    python tools/bench_gemm_vs_vendor.py [--epilogue]      --epilogue: also the step's fused epilogues on our kernels
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402
from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched  # noqa: E402

BF = torch.bfloat16
VARIANTS = {          # name -> tuning overrides
    "default": {},
    "128^2 KD32": {"GEMM_GLDS_FORCE": 1, "GEMM_GLDS_KD": 32, "GEMM_P8_MINK": 1 << 30},
    "128^2 KD64": {"GEMM_GLDS_FORCE": 1, "GEMM_GLDS_KD": 64, "GEMM_P8_MINK": 1 << 30},
    "256^2 p8": {"GEMM_P8": 1, "GEMM_GLDS_FORCE": 1},
}


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--epilogue", action="store_true")
    ap.add_argument("--shapes", default="c5")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L.set_gemm_precision("bf16")
    shapes = [("C5 fc1 y=xW^T", 16200, 3072, 768, "nt"), ("C5 fc2 y=hW^T", 16200, 768, 3072, "nt"), ("C5 gh=gW2", 16200, 3072, 768, "nn"),
              ("C5 gx=ghW1", 16200, 768, 3072, "nn"), ("C5' fc1 (721)", 18540, 3072, 768, "nt"), ("C5' fc2 (721)", 18540, 768, 3072, "nt")]
    if a.shapes == "all":
        shapes += [("Pangu fc1 L2", 8192, 1536, 384, "nt"), ("Pangu fc2 L2", 8192, 384, 1536, "nt"), ("Pangu fc1 L1", 32768, 768, 192, "nt"),
                   ("Swin fc1", 65536, 384, 96, "nt"), ("sq 4096", 4096, 4096, 4096, "nt"), ("sq 8192", 8192, 8192, 8192, "nt")]
    g = torch.Generator().manual_seed(0)
    for name, M, N, K, lay in shapes:
        A = (torch.rand(M, K, generator=g) * 2 - 1).to(dev).to(BF)
        B = (torch.rand((N, K) if lay[1] == "t" else (K, N), generator=g) * 2 - 1).to(dev).to(BF)
        Y = torch.empty(M, N, device=dev, dtype=BF)
        Z = torch.empty(M, N, device=dev, dtype=BF)
        bias = torch.randn(N, generator=g).to(dev)
        Bm = B.t() if lay[1] == "t" else B
        tB, ldb = (1, K) if lay[1] == "t" else (0, N)
        fns = {"vendor (torch.matmul)": lambda: torch.matmul(A, Bm, out=Y)}
        for vn in VARIANTS:
            fns[vn] = lambda: _gemm(A, B, Y, M, N, K, K, ldb, N, 0, tB)
        if a.epilogue and lay == "nt" and N > K:
            for vn in VARIANTS:
                fns[vn + " +bias+GELU+z"] = lambda: _gemm(A, B, Y, M, N, K, K, ldb, N, 0, tB, bias, 7, Z, None)
        if a.epilogue and lay == "nn" and N > K:
            for vn in VARIANTS:
                fns[vn + " *stored GELU'"] = lambda: _gemm_batched(A, B, Y, M, N, K, K, ldb, N, 0, tB, act=8, residual=Z)
        res = {k: [] for k in fns}
        ref = None
        for rnd in range(a.rounds + 1):
            for k, fn in fns.items():
                base = k.split(" +")[0].split(" *")[0]
                for kn, kv in VARIANTS.get(base, {}).items():
                    L.set_tuning(kn, kv)
                try:
                    us = timed(fn, a.reps)
                finally:
                    for kn in VARIANTS.get(base, {}):
                        L.set_tuning(kn, None)
                if rnd:
                    res[k].append(us)
                elif "+" not in k and "*" not in k:          # first round: warm-up + a value check of every plain variant
                    if ref is None:
                        ref = Y.float().clone()
                    else:
                        err = ((Y.float() - ref).abs().max() / ref.abs().max()).item()
                        assert err < 2e-2, (name, k, err)
        fl = 2.0 * M * N * K
        print(f"== {name}  M={M} N={N} K={K} {lay}")
        for k, v in res.items():
            v.sort()
            med, mn = v[len(v) // 2], v[0]
            print(f"   {k:34s} median {med:7.1f} us {fl / med / 1e6:7.1f} TF   min {mn:7.1f} us {fl / mn / 1e6:7.1f} TF", flush=True)


if __name__ == "__main__":
    main()
