#!/usr/bin/env python3
"""Phase stamps of winattn_lds_bwd1p_kernel (workgroup 0, first window; needs `make stamps`): cycles on the s_memtime clock."""
import ctypes as C, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
os.environ["DLWP_LIB_FILE"] = "libdlwpmi_stamps.so"
sys.path.insert(0, os.path.join(HERE, ".."))
import torch
from dlwp_benchmark_amd import lib as L
lib = L.load()
raw = C.CDLL(L.LIB_PATH)
raw.dlwp_debug_stamps_winattn_small.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
L.set_gemm_precision("bf16")
B_, nW, N, heads, d, TB, ntypes, qr = 703, 703, 98, 6, 32, 2548, 19, (49, 98)
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
for _ in range(3):
    L.check(lib.dlwp_window_attn_fwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out), L.ptr(lse),
                                            B_, nW, N, TB, ntypes, heads, d, d ** -0.5, qr[0], qr[1], st))
    L.check(lib.dlwp_window_attn_bwd_qrange(L.ptr(qkv), L.ptr(table), L.ptr(packed), L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out), L.ptr(lse),
                                            L.ptr(gout), L.ptr(gqkv), L.ptr(gtable), L.ptr(dsum), L.ptr(slab), B_, nW, N, TB, ntypes, heads, d,
                                            d ** -0.5, qr[0], qr[1], st))
    torch.cuda.synchronize()
buf = (C.c_ulonglong * 32)()
raw.dlwp_debug_stamps_winattn_small(buf)
t = list(buf)
print("init", t[1] - t[0], "| window 1: stage", t[2] - t[1], "pass1 (to barrier)", t[3] - t[2], "waves end pass1 at", [t[8 + i] - t[2] for i in range(8)])
print("wave 0 pass-1 steps begin at", [t[24 + i] - t[2] for i in range(4)])
print("whole window loop", t[4] - t[1], "| last waves reach the end at", [t[16 + i] - t[1] for i in range(8)], "| fold", t[5] - t[4], "flush", t[6] - t[5], "| total", t[6] - t[0])
