#!/usr/bin/env python3
"""Phase stamps of gemm_glds_kernel's workgroup 0 (needs `make stamps`) at the Pangu / Swin C4 product shapes, y = x W^T ("nt") and
gx = g W ("nn"): first K-step | remaining K-steps | epilogue in s_memtime ticks (10 ns), next to the event-timed launch, for the
tile-width / ring-depth variants (DLWP_GEMM_GLDS_N96, DLWP_GEMM_GLDS_STAGES are read at every call)."""
import ctypes as C
import os

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "dlwp_benchmark_amd", "libdlwpmi_stamps.so"))
V, I = C.c_void_p, C.c_int
lib.dlwp_gemm_mixed.argtypes = [V, V, V] + [I] * 8 + [V, I, V, V, I, V, I, V]
lib.dlwp_set_gemm_precision.argtypes = [I]
lib.dlwp_debug_stamps_gemm.argtypes = [V]
lib.dlwp_set_tuning.argtypes = [C.c_char_p, I]
lib.dlwp_set_gemm_precision(1)
dev = "cuda"
lib.dlwp_set_tuning(b"GEMM_GLDS_FORCE", 1)
lib.dlwp_set_tuning(b"GEMM_GLDS_STAGES_MAXTILES", 1 << 30)
for (M, N, K, form) in [(8192, 384, 1536, "nt"), (8192, 384, 1536, "nn"), (8192, 384, 384, "nt"), (8192, 384, 384, "nn"), (8192, 1536, 384, "nt"),
                        (8192, 1536, 384, "nn")]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) if form == "nt" else torch.randn(K, N, device=dev)).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for n96, st in [(0, 2), (2, 2), (2, 4), (0, 4)]:
        lib.dlwp_set_tuning(b"GEMM_GLDS_N96", n96)
        lib.dlwp_set_tuning(b"GEMM_GLDS_STAGES", st)

        def go():
            rc = lib.dlwp_gemm_mixed(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K if form == "nt" else N, N, 0, 1 if form == "nt" else 0,
                                     None, 0, None, None, 0, None, 7, None)
            assert rc == 0, rc
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            go()
        e1.record()
        torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        lib.dlwp_debug_stamps_gemm(buf)
        t = list(buf)
        print(f"{(M, N, K)} {form} n96={n96} stages={st}: first step {t[11] - t[10]:5d}  other steps {t[12] - t[11]:5d}  epilogue {t[13] - t[12]:5d}  "
              f"total {t[13] - t[10]:5d} ticks of 10 ns   launch {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us back to back")
