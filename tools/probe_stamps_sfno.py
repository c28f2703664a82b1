#!/usr/bin/env python3
"""Phase stamps (needs `make stamps`) and stand-alone timing of the three spectral kernels of the SFNO block at the C3 shape
(32 x 64 grid, C = 256, lmax = mmax = 32): bf16 synthesis / analysis (csrc/sht_bf16.hip) and the per-degree complex channel
mixing (csrc/dhconv.hip).  Usage: probe_stamps_sfno.py [--nostamps] [B ...]"""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
stamps = "--nostamps" not in sys.argv
lib = C.CDLL(os.path.join(here, "..", "dlwp_benchmark_amd", "libdlwpmi_stamps.so" if stamps else "libdlwpmi.so"))
V, I = C.c_void_p, C.c_int
lib.dlwp_sht_analysis_bf16.argtypes = [V, V, V, V, I, I, I, I, I, I, V]
lib.dlwp_sht_synthesis_bf16_ex.argtypes = [V, V, V, V, V, I, I, I, I, I, I, I, V]
lib.dlwp_dhconv_apply.argtypes = [V, V, V, I, I, I, I, I, I, V]
lib.dlwp_last_error.restype = C.c_char_p
lib.dlwp_set_tuning.argtypes = [C.c_char_p, I]
lib.dlwp_clear_tuning.argtypes = [C.c_char_p]
if stamps:
    lib.dlwp_debug_stamps_sht.argtypes = [V]
    lib.dlwp_debug_stamps_dhconv.argtypes = [V]
dev, BF = "cuda", torch.bfloat16
K, N, Cc, M, Lm = 32, 64, 256, 32, 32


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def stamp_line(getter, names, base):
    buf = (C.c_ulonglong * 32)()
    getter(buf)
    s = list(buf)
    return ", ".join(f"{n} {s[base + i + 1] - s[base + i]}" for i, n in enumerate(names)) + f"; total {s[base + len(names)] - s[base]}"


for B in [int(a) for a in sys.argv[1:] if a.isdigit()] or [4, 16]:
    x = torch.randn(B, K, N, Cc, device=dev)
    res = torch.randn(B, K, N, Cc, device=dev)
    out = torch.empty(B, K, N, Cc, device=dev)
    X = torch.randn(Lm, B, M, 2, Cc, device=dev).to(BF)
    tri = (torch.arange(M, device=dev)[None, :] <= torch.arange(Lm, device=dev)[:, None]).to(BF)
    X = (X * tri[:, None, :, None, None]).contiguous()
    Y = torch.empty_like(X)
    A1 = (torch.randn(2 * M, N, device=dev) / 8).to(BF)
    A2 = (torch.randn(M, Lm, K, device=dev) / 8).to(BF)
    S1t = (torch.randn(M, K, Lm, device=dev) / 8).to(BF)
    S2 = (torch.randn(N, 2 * M, device=dev) / 8).to(BF)
    img = (torch.randn(Lm * 2 * Cc * Cc, device=dev) / 16).to(BF)

    def chk(rc):
        assert rc == 0, lib.dlwp_last_error()

    syn = lambda r=None, tri=1: chk(lib.dlwp_sht_synthesis_bf16_ex(X.data_ptr(), S1t.data_ptr(), S2.data_ptr(), r, out.data_ptr(), B, K, N, Cc, M, Lm, tri, None))
    ana = lambda: chk(lib.dlwp_sht_analysis_bf16(x.data_ptr(), A1.data_ptr(), A2.data_ptr(), Y.data_ptr(), B, K, N, Cc, M, Lm, None))
    dh = lambda: chk(lib.dlwp_dhconv_apply(X.data_ptr(), img.data_ptr(), Y.data_ptr(), Lm, B * M * 2, Cc, Cc, M, 0, None))
    dhd = lambda: chk(lib.dlwp_dhconv_apply(X.data_ptr(), img.data_ptr(), Y.data_ptr(), Lm, B * M * 2, Cc, Cc, 0, 0, None))
    print(f"B={B} synthesis (all orders read) {timed(lambda: syn(None, 0)):.2f} us per launch", flush=True)
    print(f"B={B} synthesis        {timed(lambda: syn(None)):.2f} us per launch", flush=True)
    print(f"B={B} synthesis + res  {timed(lambda: syn(res.data_ptr())):.2f} us per launch", flush=True)
    print(f"B={B} analysis         {timed(ana):.2f} us per launch", flush=True)
    for name, knobs in (("pipelined 128", {}), ("pipelined 64", {b"DHCONV_RC": 64}), ("round 4", {b"DHCONV_APPLY": 1})):
        for k_, v_ in knobs.items():
            lib.dlwp_set_tuning(k_, v_)
        print(f"B={B} dhconv {name:14s} sparse {timed(dh):.2f} us, dense {timed(dhd):.2f} us per launch", flush=True)
        if stamps:
            dh(); torch.cuda.synchronize()
            nm = ["dma issue", "wload+vmcnt0", "barrier", "mma+stores", "drain"] if b"DHCONV_APPLY" in knobs else ["W + chunk 0 issue", "wait + barrier", "chunk walk", "drain"]
            print("    heaviest degree, cycles:", stamp_line(lib.dlwp_debug_stamps_dhconv, nm, 0))
        for k_ in knobs:
            lib.dlwp_clear_tuning(k_)
    if stamps:
        syn(res.data_ptr()); torch.cuda.synchronize()
        print("  synthesis+res wg0 cycles:", stamp_line(lib.dlwp_debug_stamps_sht, ["issue", "vmcnt0", "barrier", "stage1", "->barrier", "barrier", "stage2+stores"], 8))
        syn(None); torch.cuda.synchronize()
        print("  synthesis     wg0 cycles:", stamp_line(lib.dlwp_debug_stamps_sht, ["issue", "vmcnt0", "barrier", "stage1", "->barrier", "barrier", "stage2+stores"], 8))
        ana(); torch.cuda.synchronize()
        print("  analysis      wg0 cycles:", stamp_line(lib.dlwp_debug_stamps_sht, ["prologue", "stage1", "t2+barrier", "stage2", "drain"], 0))
