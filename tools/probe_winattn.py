#!/usr/bin/env python3
"""Time the window-attention kernels alone at Swin / Pangu shapes (diagnostic)."""
import sys
import torch
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from dlwp_benchmark_amd import lib as L

lib = L.load()
dev = torch.device("cuda:0")


def run(B_, nW, N, heads, d, TB, masked, iters=50, ntypes=1):
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B_, N, 3, heads, d, generator=g).to(dev)
    table = (torch.randn(TB, ntypes, heads, generator=g) * 0.02).to(dev)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(dev) if masked else None
    out = torch.empty(B_, N, heads * d, device=dev)
    lse = torch.empty(B_, heads, N, device=dev)
    gout = torch.randn_like(out)
    gqkv = torch.empty_like(qkv)
    gtable = torch.zeros_like(table)
    dsum = torch.empty_like(lse)
    slab = torch.empty(lib.dlwp_window_attn_bwd_slab_floats(B_, N, heads, TB), device=dev)
    st = torch.cuda.current_stream().cuda_stream

    packed = torch.empty(ntypes * heads * TB, device=dev) if ntypes > 1 else None

    def fwd():
        if packed is not None:
            L.check(lib.dlwp_window_attn_pack_table(L.ptr(table), L.ptr(packed), TB, ntypes, heads, st))
        L.check(lib.dlwp_window_attn_fwd_packed(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out), L.ptr(lse),
                                         B_, nW, N, TB, ntypes, heads, d, d ** -0.5, st))

    def bwd():
        L.check(lib.dlwp_window_attn_bwd_packed(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out), L.ptr(lse),
                                         L.ptr(gout), L.ptr(gqkv), L.ptr(gtable), L.ptr(dsum), L.ptr(slab),
                                         B_, nW, N, TB, ntypes, heads, d, d ** -0.5, st))
    res = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    print(f"B_={B_:5d} nW={nW:3d} N={N:5d} heads={heads} d={d:3d} TB={TB:5d} ntypes={ntypes} masked={int(masked)}: fwd {res[0]:8.1f} us  bwd {res[1]:8.1f} us")


run(1, 1, 49, 4, 10, 169, False)
run(100, 25, 49, 4, 10, 169, False)
run(100, 25, 49, 4, 10, 169, True)
run(400, 25, 49, 4, 10, 169, True)
run(36, 9, 49, 4, 20, 169, True)
run(64, 16, 64, 4, 16, 225, True)
run(16, 16, 144, 6, 32, 3312, True)     # Pangu-like window (2,6,12)
run(4, 1, 2048, 4, 24, 8001, True, iters=10)   # dlwpbench Swin default: whole 32x64 map per window

print("# C4 shapes")
run(50, 50, 98, 6, 32, 2548, True, ntypes=5)      # Pangu C4 layer 1 / 4: (2,7,7) windows of the padded 2 x 35 x 70 grid
run(50, 50, 98, 6, 32, 2548, True, ntypes=1)
run(50, 50, 98, 6, 32, 169, True, ntypes=1)
run(15, 15, 98, 12, 32, 2548, True, ntypes=3)     # Pangu C4 layers 2 / 3
run(1406, 703, 49, 4, 24, 169, True)              # dlwpbench Swin C4, stage 1 (B = 2)
run(360, 180, 49, 4, 48, 169, True)               # stage 2
run(703, 703, 98, 6, 32, 2548, True, ntypes=19, iters=20)    # Pangu C4 at patch 1 (bench_models pangu_c4), layers 1 / 4
run(180, 180, 98, 12, 32, 2548, True, ntypes=10, iters=20)   # layers 2 / 3 (2 x 70 x 133 padded -> 10 x 19 windows ... 180)
