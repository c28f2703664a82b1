"""Un-profiled per-launch times (HIP events over back-to-back launches; eager and torch CUDA graph)."""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import lib as L
lib = L.load(); dev = 'cuda'
s = torch.cuda.Stream()

def timeit(fn, n=200):
    with torch.cuda.stream(s):
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(n): fn()
        e1.record(s); torch.cuda.synchronize()
        eager = e0.elapsed_time(e1) * 1e3 / n
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
        g.replay(); torch.cuda.synchronize()
        e0.record(s); g.replay(); e1.record(s); torch.cuda.synchronize()
        return eager, e0.elapsed_time(e1) * 1e3 / n

B, Cc, H, W, m1, m2c = 4, 32, 64, 64, 12, 7
plan = C.c_void_p(); L.check(lib.dlwp_fno_plan_create(Cc, H, W, m1, m2c, C.byref(plan)))
ws = torch.empty(lib.dlwp_fno_block_workspace_bytes(plan, B), dtype=torch.uint8, device=dev)
x = torch.randn(B, Cc, H, W, device=dev); w = torch.randn(m1, m2c, Cc, Cc, 2, device=dev)
k = torch.randn(Cc, Cc, device=dev); bb = torch.randn(Cc, device=dev)
pre = torch.empty_like(x); xhat = torch.empty(B, m1, m2c, Cc, 2, device=dev); g_ = torch.randn_like(x); gx = torch.empty_like(x)
gw = torch.zeros_like(w); gk = torch.zeros_like(k); gb = torch.zeros_like(bb)
st = lambda: s.cuda_stream
print("block_fwd (rows+mix+spatial) us eager/graph:", timeit(lambda: L.check(lib.dlwp_fno_block_fwd(plan, L.ptr(x), 1, L.ptr(w), L.ptr(k), L.ptr(bb), L.ptr(pre), L.ptr(xhat), B, L.ptr(ws), st()))))
print("block_bwd (rows+mix+spatial) us eager/graph:", timeit(lambda: L.check(lib.dlwp_fno_block_bwd(plan, L.ptr(x), 1, L.ptr(w), L.ptr(k), L.ptr(g_), L.ptr(xhat), L.ptr(gx), L.ptr(gw), L.ptr(gk), L.ptr(gb), B, L.ptr(ws), st()))))
for (Cin, Ch, Cout) in [(10, 256, 32), (32, 256, 1)]:
    P = H * W
    xx = torch.randn(B, Cin, P, device=dev); w1 = torch.randn(Ch, Cin, device=dev); b1 = torch.randn(Ch, device=dev)
    w2 = torch.randn(Cout, Ch, device=dev); b2 = torch.randn(Cout, device=dev); y = torch.empty(B, Cout, P, device=dev)
    gy = torch.randn(B, Cout, P, device=dev); gxx = torch.empty_like(xx)
    gg = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]
    print(f"pwmlp_fwd {Cin}->{Ch}->{Cout} us eager/graph:", timeit(lambda: L.check(lib.dlwp_pwmlp_fwd(L.ptr(xx), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(b2), L.ptr(y), B, Cin, Ch, Cout, P, st()))))
    print(f"pwmlp_bwd {Cin}->{Ch}->{Cout} (atomics path) us eager/graph:", timeit(lambda: L.check(lib.dlwp_pwmlp_bwd(L.ptr(xx), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(gy), L.ptr(gxx), *[L.ptr(t) for t in gg], B, Cin, Ch, Cout, P, st()))))
print("null x1 us eager/graph:", timeit(lambda: L.check(lib.dlwp_debug_null_kernels(1, 256, st()))))
