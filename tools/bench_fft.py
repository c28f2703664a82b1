"""rFFT2 / irFFT2 kernel rates and the AFNO2D path crossover (dense DFT GEMMs vs FFT).  HIP-event timing on the launch
stream.  TWO byte conventions per transform: "in+out" = the minimum any implementation moves (read the real field once, write
the half spectrum once) and "two-pass" = what this implementation's two launches move (W axis: real field in, half spectrum
out; H axis: half spectrum in and out).  Fractions are of the 8 TB/s HBM roof.
    python tools/bench_fft.py > gpurun_out/r03_fft_bench.txt      (from any directory)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import fft  # noqa: E402
from dlwp_benchmark_amd.nsbench.fourcastnet import AFNO2D  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


print("# transform rates (channels-last, ortho)")
for (B, H, W, C) in [(4, 64, 64, 64), (4, 32, 64, 64), (4, 128, 256, 64), (2, 128, 256, 256), (1, 721, 1440, 64), (1, 720, 1440, 64),
                     (1, 103, 180, 768)]:
    x = torch.randn(B, H, W, C, device=dev)
    Wc = W // 2 + 1
    X = fft.rfft2(x)
    tf = timeit(lambda: fft.rfft2(x))
    ti = timeit(lambda: fft.irfft2(X, W))
    real_b, spec_b = 4.0 * B * H * W * C, 8.0 * B * H * Wc * C
    bytes_min, bytes_2p = real_b + spec_b, real_b + 3 * spec_b
    print(f"rfft2  {B}x{H}x{W}x{C}: {tf * 1e6:9.1f} us  in+out {bytes_min / tf / 1e9:7.1f} GB/s ({bytes_min / tf / 8e12:.3f} of HBM)  "
          f"two-pass {bytes_2p / tf / 1e9:7.1f} GB/s ({bytes_2p / tf / 8e12:.3f})   "
          f"irfft2: {ti * 1e6:9.1f} us  in+out {bytes_min / ti / 1e9:7.1f} GB/s ({bytes_min / ti / 8e12:.3f})  "
          f"two-pass {bytes_2p / ti / 1e9:7.1f} GB/s ({bytes_2p / ti / 8e12:.3f})", flush=True)
    # torch.fft (rocFFT) as a comparator, same layout
    tt = timeit(lambda: torch.fft.rfft2(x, dim=(1, 2), norm="ortho"))
    print(f"       torch.fft.rfft2 (rocFFT) comparator: {tt * 1e6:9.1f} us", flush=True)

print("# AFNO2D forward + backward, dense-DFT GEMM path vs FFT path")
for (B, H, W, C, nb) in [(4, 32, 64, 64, 4), (4, 64, 64, 64, 4), (2, 64, 128, 64, 4), (2, 128, 256, 64, 4), (1, 90, 180, 768, 16),
                         (1, 103, 180, 768, 16), (1, 256, 512, 64, 4), (1, 721, 1440, 64, 4)]:
    m = AFNO2D(C, num_blocks=nb).to(dev)
    x = torch.randn(B, H, W, C, device=dev, requires_grad=True)
    res = {}
    for path in ("tiled", "fft"):
        m.path = path
        if path == "tiled" and H * W > 300000:
            res[path] = float("nan")       # hours of dense DFT: skipped
            continue

        def step():
            y = m(x)
            y.sum().backward()
        try:
            res[path] = timeit(step, reps=5, warm=2)
        except Exception as exc:       # noqa: BLE001
            res[path] = float("nan")
            print("  ", path, "failed:", str(exc)[:100])
    print(f"afno2d {B}x{H}x{W}x{C}: tiled {res['tiled'] * 1e3:9.3f} ms   fft {res['fft'] * 1e3:9.3f} ms", flush=True)
