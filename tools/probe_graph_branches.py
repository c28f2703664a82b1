#!/usr/bin/env python3
"""Do two independent kernel chains captured as parallel branches of ONE hipGraph overlap on this runtime?
Each kernel is a small latency-bound launch (64 workgroups); chains of N kernels:
  serial: one chain of 2N;  branches: two chains of N forked / joined inside one capture;  single: one chain of N."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dlwp_benchmark_amd import lib as L  # noqa: E402

dev = torch.device("cuda:0")
lib = L.load()
N = 100
a = [torch.randn(64, 4096, device=dev) for _ in range(2)]
b = [torch.empty_like(t) for t in a]


def kernel(i, stream):
    # layernorm over 64 rows x 4096: 16 workgroups of one wave per row -> a latency-bound launch
    L.check(lib.dlwp_scale_rows_add(L.ptr(a[i]), None, L.ptr(a[i]), L.ptr(b[i]), 64, 4096, stream))


def capture(mode):
    g = torch.cuda.CUDAGraph()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if mode == "serial":
            for _ in range(2 * N):
                kernel(0, main.cuda_stream)
        elif mode == "single":
            for _ in range(N):
                kernel(0, main.cuda_stream)
        else:
            s2.wait_stream(main)
            for _ in range(N):
                kernel(0, main.cuda_stream)
            with torch.cuda.stream(s2):
                for _ in range(N):
                    kernel(1, s2.cuda_stream)
            main.wait_stream(s2)
    return g


for mode in ("single", "serial", "branches"):
    g = capture(mode)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    print(f"{mode:9s}: {dt * 1e6:8.1f} us per replay")
