#!/usr/bin/env python3
"""Phase stamps of winattn_bwd_q_kernel (needs `make stamps`)."""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V, I, F = C.c_void_p, C.c_int, C.c_float
lib.dlwp_window_attn_fwd.argtypes = [V] * 7 + [I] * 7 + [F, V]
lib.dlwp_window_attn_bwd.argtypes = [V] * 12 + [I] * 7 + [F, V]
lib.dlwp_window_attn_bwd_slab_floats.restype = C.c_longlong
lib.dlwp_window_attn_bwd_slab_floats.argtypes = [I] * 4
lib.dlwp_debug_stamps_winattn.argtypes = [V]
dev = 'cuda'
for (B_, nW, N, heads, d, TB, NT) in [(16, 16, 144, 6, 32, 3312, 1), (100, 25, 49, 4, 10, 169, 1), (703, 703, 98, 6, 32, 2548, 19),
                                     (1406, 703, 49, 4, 24, 169, 1)]:
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B_, N, 3, heads, d, generator=g).to(dev)
    table = (torch.randn(TB, NT, heads, generator=g) * 0.02).to(dev)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(dev)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(dev)
    out = torch.empty(B_, N, heads * d, device=dev); lse = torch.empty(B_, heads, N, device=dev)
    gout = torch.randn_like(out); gqkv = torch.empty_like(qkv); gtable = torch.zeros_like(table); dsum = torch.empty_like(lse)
    slab = torch.empty(lib.dlwp_window_attn_bwd_slab_floats(B_, N, heads, TB), device=dev)
    p = lambda t: t.data_ptr()
    for _ in range(3):
        assert lib.dlwp_window_attn_fwd(p(qkv), p(table), p(ia), p(ib), p(labels), p(out), p(lse), B_, nW, N, TB, NT, heads, d, d ** -0.5, None) == 0
        assert lib.dlwp_window_attn_bwd(p(qkv), p(table), p(ia), p(ib), p(labels), p(out), p(lse), p(gout), p(gqkv), p(gtable), p(dsum), p(slab),
                                        B_, nW, N, TB, NT, heads, d, d ** -0.5, None) == 0
        torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.dlwp_debug_stamps_winattn(buf)
    t = list(buf)
    print((B_, nW, N, heads, d, TB, NT), "bwd_q stamps 0..12 deltas", [t[i + 1] - t[i] for i in range(12)], "total", t[12] - t[0],
          "| prologue: issue", t[13] - t[0], "zero", t[14] - t[13], "table", t[15] - t[14], "gtb0", t[16] - t[15], "rest", t[1] - t[16])
