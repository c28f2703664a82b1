#!/usr/bin/env python3
"""Phase stamps of gemm_glds_kernel's workgroup 0 (needs `make stamps`): prologue + first K-step | remaining K-steps | epilogue,
in s_memtime ticks (100 MHz), for y = x W^T with both operands bf16 arrays."""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(here, "..", "dlwp_benchmark_amd", "libdlwpmi_stamps.so"))
V, I = C.c_void_p, C.c_int
lib.dlwp_gemm_mixed.argtypes = [V, V, V] + [I] * 8 + [V, I, V, V, I, V, I, V]
lib.dlwp_set_gemm_precision.argtypes = [I]
lib.dlwp_debug_stamps_gemm.argtypes = [V]
lib.dlwp_set_gemm_tile256.argtypes = [I]
lib.dlwp_set_gemm_precision(1)
dev = "cuda"
for (M, N, K) in [(16200, 3072, 768), (16200, 768, 3072), (4096, 4096, 4096)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    w = torch.randn(N, K, device=dev).bfloat16()
    y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        rc = lib.dlwp_gemm_mixed(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, 0, 1, None, 0, None, None, 0, None, 7, None)
        assert rc == 0, rc
        torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.dlwp_debug_stamps_gemm(buf)
    t = list(buf)
    print((M, N, K), "128^2 kernel: first step", t[11] - t[10], "other steps", t[12] - t[11], "epilogue", t[13] - t[12], "total", t[13] - t[10], "cycles")
    if hasattr(lib, "dlwp_set_gemm_tile256"):
        lib.dlwp_set_gemm_tile256(1)
        for _ in range(3):
            rc = lib.dlwp_gemm_mixed(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, 0, 1, None, 0, None, None, 0, None, 7, None)
            assert rc == 0, rc
            torch.cuda.synchronize()
        lib.dlwp_debug_stamps_gemm(buf)
        t = list(buf)
        print((M, N, K), "256^2 kernel (first tile of workgroup 0): prologue", t[15] - t[14], "K loop", t[16] - t[15], "epilogue", t[17] - t[16], "cycles")
        lib.dlwp_set_gemm_tile256(0)
