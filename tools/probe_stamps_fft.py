#!/usr/bin/env python3
"""Phase stamps (needs `make stamps`) and stand-alone time of the rFFT2 / irFFT2 passes at the C5 token grid (1 x 90 x 180 x 768,
channels-last, ortho): W-axis real -> complex, H-axis complex -> complex, and back.  Usage: probe_stamps_fft.py [--nostamps] [H W C]"""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
stamps = "--nostamps" not in sys.argv
lib = C.CDLL(os.path.join(here, "..", "dlwp_benchmark_amd", "libdlwpmi_stamps.so" if stamps else "libdlwpmi.so"))
V, I = C.c_void_p, C.c_int
lib.dlwp_fft_plan_create.argtypes = [I, I, C.POINTER(V)]
lib.dlwp_rfft2.argtypes = [V, V, V, I, I, I, I, I, V]
lib.dlwp_irfft2.argtypes = [V, V, V, V, I, I, I, I, I, V]
lib.dlwp_last_error.restype = C.c_char_p
dims = [int(a) for a in sys.argv[1:] if a.isdigit()]
H, W, Cc = dims if len(dims) == 3 else (90, 180, 768)
B = 1
plan = V()
assert lib.dlwp_fft_plan_create(H, W, C.byref(plan)) == 0, lib.dlwp_last_error()
x = torch.randn(B, H, W, Cc, device="cuda")
X = torch.empty(B, H, W // 2 + 1, Cc, 2, device="cuda")
work = torch.empty_like(X)
y = torch.empty_like(x)
st = torch.cuda.current_stream().cuda_stream


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


def fwd():
    rc = lib.dlwp_rfft2(plan, x.data_ptr(), X.data_ptr(), B, Cc, 0, 0, 0, st)
    assert rc == 0, lib.dlwp_last_error()


def inv():
    rc = lib.dlwp_irfft2(plan, X.data_ptr(), y.data_ptr(), work.data_ptr(), B, Cc, 0, 0, 0, st)
    assert rc == 0, lib.dlwp_last_error()


nbytes = x.numel() * 4 + X.numel() * 4
print(f"rfft2 {B}x{H}x{W}x{Cc}: {timed(fwd):7.1f} us   irfft2: {timed(inv):7.1f} us   (field + spectrum = {nbytes / 1e6:.1f} MB per pass pair)")
if stamps:
    lib.dlwp_debug_stamps_fft.argtypes = [V]
    for name, fn in (("rfft2", fwd), ("irfft2", inv)):
        fn()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        lib.dlwp_debug_stamps_fft(buf)
        s = list(buf)
        for tag, b0 in (("r2c", 0), ("c2c", 4), ("c2r", 8)):
            if s[b0 + 3] > s[b0] > 0:
                print(f"  {name} {tag}: load {s[b0 + 1] - s[b0]} | passes {s[b0 + 2] - s[b0 + 1]} | store {s[b0 + 3] - s[b0 + 2]} cycles (workgroup 0, thread 0)")
        print("  last kernel's pass ends relative to its first:", [s[16 + i] - s[16] for i in range(6) if s[16 + i] >= s[16]])
