"""A few rFFT2 / irFFT2 launches for rocprofv3 (kernel durations of the FFT stages; bytes per launch: tools/bench_fft.py)."""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import fft
dev = torch.device("cuda:0")
for (B, H, W, C) in [(4, 128, 256, 64), (2, 128, 256, 256), (1, 720, 1440, 64), (1, 721, 1440, 64)]:
    x = torch.randn(B, H, W, C, device=dev)
    for _ in range(12):
        X = fft.rfft2(x)
        y = fft.irfft2(X, W)
    torch.cuda.synchronize()
    print(B, H, W, C, "ok")
