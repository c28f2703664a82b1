"""Probe: per-kernel floor of dependent launches on this GPU (eager stream vs torch CUDA graph)."""
import sys, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L
lib = L.load()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for blocks in (1, 256, 1024):
        for n in (1000,):
            L.check(lib.dlwp_debug_null_kernels(50, blocks, s.cuda_stream)); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s); L.check(lib.dlwp_debug_null_kernels(n, blocks, s.cuda_stream)); e1.record(s)
            torch.cuda.synchronize()
            print(f"eager blocks={blocks} per-kernel {e0.elapsed_time(e1)*1e3/n:.2f} us")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                L.check(lib.dlwp_debug_null_kernels(n, blocks, s.cuda_stream))
            g.replay(); torch.cuda.synchronize()
            e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
            print(f"graph blocks={blocks} per-kernel {e0.elapsed_time(e1)*1e3/n:.2f} us")
