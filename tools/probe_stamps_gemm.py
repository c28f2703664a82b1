#!/usr/bin/env python3
"""Phase stamps of gemm_kernel (needs `make stamps`): prologue | k-step 0 | k-step 1 | remaining k-steps | epilogue."""
import ctypes as C, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V, I = C.c_void_p, C.c_int
lib.dlwp_gemm.argtypes = [V, V, V] + [I] * 8 + [V, I, V, V, I, V, V]
lib.dlwp_set_gemm_precision.argtypes = [I]
lib.dlwp_debug_stamps_gemm.argtypes = [V]
dev = 'cuda'
for prec in (0, 1):
    lib.dlwp_set_gemm_precision(prec)
    for (M, N, K) in [(8192, 512, 256), (32768, 512, 256), (5184, 576, 192)]:
        x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev); b = torch.zeros(N, device=dev)
        y = torch.empty(M, N, device=dev); z = torch.empty(M, N, device=dev)
        for _ in range(3):
            assert lib.dlwp_gemm(x.data_ptr(), w.data_ptr(), y.data_ptr(), M, N, K, K, K, N, 0, 1, b.data_ptr(), 1, z.data_ptr(), None, 0, None, None) == 0
            torch.cuda.synchronize()
        buf = (C.c_ulonglong * 32)()
        lib.dlwp_debug_stamps_gemm(buf)
        t = list(buf)
        print("bf16" if prec else "fp32", (M, N, K), "deltas", [t[i + 1] - t[i] for i in range(5)], "total", t[5] - t[0])
