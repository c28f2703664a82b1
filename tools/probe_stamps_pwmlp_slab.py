"""Phase stamps of pwmlp_bwd_kernel in slab mode (the path the rollout trainer uses); needs `make stamps`."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V = C.c_void_p; I = C.c_int
lib.dlwp_pwmlp_bwd_slab.argtypes = [V] * 7 + [I] * 6 + [V]; lib.dlwp_pwmlp_bwd_slab.restype = I
lib.dlwp_pwmlp_slab_floats.argtypes = [I] * 5; lib.dlwp_pwmlp_slab_floats.restype = C.c_longlong
lib.dlwp_debug_stamps_pwmlp.argtypes = [V]
dev = 'cuda'
names = ["10 setup", "11", "12 first-iter start", "13 z", "14 gat", "15 gelu", "16 dW", "17 dX", "18 loop rest (hb0)", "19 flush hb0", "20 other hbs", "21 red store", "22 sync"]
for (B, Cin, Ch, Cout, P) in [(4, 10, 256, 32, 4096), (4, 32, 256, 1, 4096)]:
    x = torch.randn(B, Cin, P, device=dev); w1 = torch.randn(Ch, Cin, device=dev); b1 = torch.randn(Ch, device=dev)
    w2 = torch.randn(Cout, Ch, device=dev); gy = torch.randn(B, Cout, P, device=dev); gx = torch.empty_like(x)
    slab = torch.zeros(lib.dlwp_pwmlp_slab_floats(B, Cin, Ch, Cout, P), device=dev)
    for it in range(3):
        rc = lib.dlwp_pwmlp_bwd_slab(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), gy.data_ptr(), gx.data_ptr(),
                                     slab.data_ptr(), 1, B, Cin, Ch, Cout, P, None)
        assert rc == 0
        torch.cuda.synchronize()
    buf = (C.c_ulonglong * 32)()
    lib.dlwp_debug_stamps_pwmlp(buf)
    t = list(buf)
    print("bwd slab", (B, Cin, Ch, Cout, P), "stamps 10..22 deltas:", [t[i + 1] - t[i] for i in range(10, 22)], "total", t[22] - t[10])
    print("   end of main loop per wave, cycles after stamp 10:", [t[24 + w] - t[10] for w in range(8)])
