#!/usr/bin/env python3
"""LayerNorm forward at the C4 / C5 token shapes in the form the steps use (bf16 output), back to back on one set of buffers ("warm") and
rotating over 600 MB of buffers ("cold")."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402

SHAPES = [(65536, 96, "Swin stage 1 (B=2)"), (16384, 192, "Swin stage 2"), (32768, 192, "Pangu layer 1"), (8192, 384, "Pangu layer 2"), (16200, 768, "C5 AFNO")]


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    for T, C, label in SHAPES:
        per_set = T * C * 6
        nsets = max(2, (600 << 20) // per_set + 1)
        sets = [(torch.randn(T, C, device=dev), torch.empty(T, C, device=dev, dtype=torch.bfloat16), torch.empty(T, device=dev), torch.empty(T, device=dev))
                for _ in range(nsets)]
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)

        def go(s):
            x, y, mean, rstd = s
            L.check(lib.dlwp_layernorm_fwd_ex(L.ptr(x), L.ptr(gam), L.ptr(bet), L.ptr(y), L.ptr(mean), L.ptr(rstd), T, C, 1e-5, 1, L.stream()))
        for mode in ("warm", "cold"):
            ts = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                go(sets[0])
                torch.cuda.synchronize()
                a.record()
                reps = 4 * nsets if mode == "cold" else 40
                for i in range(reps):
                    go(sets[i % nsets] if mode == "cold" else sets[0])
                b.record()
                torch.cuda.synchronize()
                ts.append(a.elapsed_time(b) * 1e3 / reps)
            t = sorted(ts)[2]
            print(f"{label:26s} {T:6d} x {C:4d} {mode}: {t:7.1f} us   {per_set / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
