#!/usr/bin/env python3
"""LayerNorm forward / backward timing at the C3-C5 token shapes (HIP events over 50 back-to-back launches, median of 5), with the
HBM floor beside it.  Environment knobs of csrc/token_ops.hip select the kernel: DLWP_LN_BWD_NOWIDE, DLWP_LN_BWD_WGS.

    python tools/probe_layernorm.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402

SHAPES = [(16200, 768, "C5 AFNO"), (32768, 192, "C4 Pangu layer 1"), (8192, 384, "C4 Pangu layer 2"), (32768, 96, "C4 Swin stage 1"),
          (8192, 256, "C3 SFNO B4")]


def timeit(fn, reps=50):
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return sorted(ts)[2]


def main():
    dev = torch.device("cuda:0")
    lib = L.load()
    for T, C, label in SHAPES:
        x, gy, ga = (torch.randn(T, C, device=dev) for _ in range(3))
        gam, gg, gb = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        mean, rstd = x.mean(1).contiguous(), (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
        gx = torch.empty_like(x)
        for name, dt in (("fwd fp32 out", torch.float32), ("fwd bf16 out", torch.bfloat16)):
            y = torch.empty(T, C, device=dev, dtype=dt)
            t = timeit(lambda: L.check(lib.dlwp_layernorm_fwd_ex(L.ptr(x), L.ptr(gam), L.ptr(gb), L.ptr(y), L.ptr(mean), L.ptr(rstd), T, C,
                                                                  1e-5, int(dt == torch.bfloat16), L.stream())))
            nbytes = T * C * (4 + y.element_size())
            print(f"{label:18s} {T:6d} x {C:4d} {name:24s} {t:7.1f} us   {nbytes / t / 1e6:6.2f} TB/s")
        for name, gadd in (("bwd", None), ("bwd + residual gradient", ga)):
            t = timeit(lambda: L.check(lib.dlwp_layernorm_bwd_ex(L.ptr(x), L.ptr(gam), L.ptr(mean), L.ptr(rstd), L.ptr(gy), 0,
                                                                  L.ptr(gadd) if gadd is not None else None, L.ptr(gx), L.ptr(gg),
                                                                  L.ptr(gb), T, C, L.stream())))
            nbytes = T * C * 4 * (3 + (gadd is not None))
            print(f"{label:18s} {T:6d} x {C:4d} {name:24s} {t:7.1f} us   {nbytes / t / 1e6:6.2f} TB/s")


if __name__ == "__main__":
    main()
