#!/usr/bin/env python3
"""Experiment: the B=4 headline step is latency-bound (one wave of 84-256 workgroups per kernel).  How much is gained by
running the samples as K independent kernel chains on K streams (same total work)?  Timing only: each chain has its own
model copy here."""
import sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
import bench
from dlwp_benchmark_amd import nsbench

w = bench.WORKLOAD
dev = torch.device("cuda:0")


def make():
    return nsbench.TFNO2DModule(n_modes=w["n_modes"], in_channels=w["in_channels"], hidden_channels=w["hidden_channels"],
                                lifting_channels=w["lifting_channels"], projection_channels=w["projection_channels"],
                                out_channels=w["out_channels"], n_layers=w["n_layers"], context_size=w["context_size"]).to(dev)


for K in (1, 2, 4):
    B = 4 // K
    models = [make() for _ in range(K)]
    streams = [torch.cuda.Stream() for _ in range(K)]
    g = torch.Generator().manual_seed(1)
    data = []
    for _ in range(K):
        u = torch.randn(B, w["T"] + 1, 1, w["H"], w["W"], generator=g).to(dev)
        data.append((u[:, :-1].contiguous(), u[:, 1:].contiguous()))

    def step():
        for m, s, (x, y) in zip(models, streams, data):
            with torch.cuda.stream(s):
                m.train_step(x, y, w["teacher_forcing_steps"], optimizer=None)
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"K={K} chains of batch {B}: {dt * 1e3:.3f} ms per 4-sample step (fwd+bwd graphs, no optimizer) -> {4 / dt:.0f} samples/s")
