"""What a dependent launch costs beyond its body, by grid and workgroup shape: chains of kernels whose waves all spin for the
same number of cycles, replayed as a graph (HIP events around the replay)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L
lib = L.load()
s = torch.cuda.Stream()
n = 500
with torch.cuda.stream(s):
    for cycles in (0, 10000):
        for (blocks, threads, lds) in [(256, 256, 4), (256, 256, 50 * 1024), (256, 512, 80 * 1024), (256, 1024, 80 * 1024),
                                       (128, 512, 100 * 1024), (64, 1024, 100 * 1024), (84, 512, 30 * 1024), (512, 256, 50 * 1024)]:
            L.check(lib.dlwp_debug_spin_kernels(20, blocks, threads, cycles, lds, s.cuda_stream)); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                L.check(lib.dlwp_debug_spin_kernels(n, blocks, threads, cycles, lds, s.cuda_stream))
            g.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            print(f"body {cycles:6d} cycles  grid {blocks:4d} x {threads:4d} threads, LDS {lds // 1024:3d} KB: {e0.elapsed_time(e1) * 1e3 / n:6.2f} us per kernel")
