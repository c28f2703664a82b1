#!/usr/bin/env python3
"""Planar-window rFFT2 / irFFT2 timing at the C5 patch grids (the AFNO2D FFT path's transforms), for sweeps of the inner-lane
counts (DLWP_FFT_IBW / DLWP_FFT_IBH, read at plan creation).   python tools/probe_fft_ib.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import fft  # noqa: E402
from dlwp_benchmark_amd.afno_tiled import kept_window  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for (B, H, W, C, bs) in [(1, 90, 180, 768, 48), (1, 103, 180, 768, 48), (1, 128, 256, 64, 16)]:
    win = kept_window(H, W, 1.0)
    x = torch.randn(B, H, W, C, device=dev)
    with torch.no_grad():
        X = fft.rfft2_planar(x, "ortho", win, block=bs)
        tf = timeit(lambda: fft.rfft2_planar(x, "ortho", win, block=bs))
        ti = timeit(lambda: fft.irfft2_planar(X, H, W, "ortho", win, block=bs))
    print(f"{B}x{H}x{W}x{C} window {win} block {bs}: rfft2_planar {tf:7.1f} us   irfft2_planar {ti:7.1f} us")
