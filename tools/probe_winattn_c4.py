#!/usr/bin/env python3
"""Window attention at the Pangu C4 shapes (layer 1: 703 windows x 6 heads, N = 98, d = 32, 19 window types; layers 2-3:
190 x 12), bf16 matrix mode, with the query range of the real-token flow; HIP-event time per launch.
Usage: DLWP_WINATTN_WG_BWD=<workgroups> python tools/probe_winattn_c4.py"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from dlwp_benchmark_amd import lib as L
lib = L.load()
dev = torch.device("cuda:0")
L.set_gemm_precision("bf16")


def run(B_, nW, N, heads, d, TB, ntypes, qr, iters=30):
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B_, N, 3, heads, d, generator=g).to(dev)
    table = (torch.randn(TB, ntypes, heads, generator=g) * 0.02).to(dev)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(dev)
    out = torch.zeros(B_, N, heads * d, device=dev); lse = torch.zeros(B_, heads, N, device=dev)
    gout = torch.randn_like(out); gqkv = torch.empty_like(qkv); gtable = torch.zeros_like(table); dsum = torch.empty_like(lse)
    slab = torch.empty(lib.dlwp_window_attn_bwd_slab_floats(B_, N, heads, TB), device=dev)
    packed = torch.empty(ntypes * heads * TB, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    L.check(lib.dlwp_window_attn_pack_table(L.ptr(table), L.ptr(packed), TB, ntypes, heads, st))
    fwd = lambda: L.check(lib.dlwp_window_attn_fwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out),
                                                          L.ptr(lse), B_, nW, N, TB, ntypes, heads, d, d ** -0.5, qr[0], qr[1], st))
    bwd = lambda: L.check(lib.dlwp_window_attn_bwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out),
                                                          L.ptr(lse), L.ptr(gout), L.ptr(gqkv), L.ptr(gtable), L.ptr(dsum), L.ptr(slab), B_, nW, N, TB,
                                                          ntypes, heads, d, d ** -0.5, qr[0], qr[1], st))
    res = []
    for f in (fwd, bwd):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / iters * 1e3)
    print(f"B_={B_} N={N} heads={heads} d={d} types={ntypes} qrange={qr}: fwd {res[0]:7.1f} us  bwd {res[1]:7.1f} us   "
          f"[WG_BWD={os.environ.get('DLWP_WINATTN_WG_BWD', '-')} NOLDS={os.environ.get('DLWP_WINATTN_NOLDS', '-')} DBG={os.environ.get('DLWP_WINATTN_DBG', '-')}]", flush=True)


SHAPES = [(703, 703, 98, 6, 32, 2548, 19, (49, 98)), (190, 190, 98, 12, 32, 2548, 10, (49, 98)), (703, 703, 98, 6, 32, 2548, 19, (0, 98))]
for i, sh in enumerate(SHAPES):
    if os.environ.get("PROBE_FIRST") and i:
        break
    run(*sh)
