#!/usr/bin/env python3
"""LayerNorm backward at the C4 token shapes in the form the step uses (gy bf16, residual gradient added), back to back on ONE set of
buffers (Infinity-Cache resident when it fits) and rotating over enough sets to exceed the 256 MB cache ("cold": what the step sees).

    python tools/probe_layernorm_bwd.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402

SHAPES = [(65536, 96, "Swin stage 1 (B=2)"), (65536, 128, "(C = 128: no idle lanes)"), (16384, 192, "Swin stage 2"), (32768, 192, "Pangu layer 1"),
          (8192, 384, "Pangu layer 2"), (16200, 768, "C5 AFNO")]


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    for T, C, label in SHAPES:
        per_set = T * C * 14
        nsets = max(2, (600 << 20) // per_set + 1)
        sets = []
        for _ in range(nsets):
            x, ga = torch.randn(T, C, device=dev), torch.randn(T, C, device=dev)
            gy = torch.randn(T, C, device=dev).bfloat16()
            sets.append((x, gy, ga, torch.empty_like(x), x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()))
        gam, gg, gb = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)

        def go(s):
            x, gy, ga, gx, mean, rstd = s
            L.check(lib.dlwp_layernorm_bwd_ex(L.ptr(x), L.ptr(gam), L.ptr(mean), L.ptr(rstd), L.ptr(gy), 1, L.ptr(ga), L.ptr(gx), L.ptr(gg), L.ptr(gb),
                                              T, C, L.stream()))
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
