"""Phase stamps of afno2d_kernel at the nsbench shape (needs `make stamps`):
tables | row DFT | column DFT | mixer | zero + parameter-gradient flush | inverse column | inverse row."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V, I, F, LL = C.c_void_p, C.c_int, C.c_float, C.c_longlong
lib.dlwp_afno2d_save_elems.argtypes = [I] * 5 + [F]; lib.dlwp_afno2d_save_elems.restype = LL
lib.dlwp_afno2d_fwd.argtypes = [V] * 7 + [I] * 5 + [F, F, V]
lib.dlwp_afno2d_bwd.argtypes = [V] * 11 + [I] * 5 + [F, F, V]
lib.dlwp_debug_stamps_afno.argtypes = [V]
dev = 'cuda'
B, H, W, Cc, nb = 4, 16, 16, 64, 4
bs = Cc // nb
x = torch.randn(B, H, W, Cc, device=dev); y = torch.empty_like(x); gy = torch.randn_like(x); gx = torch.empty_like(x)
w1 = torch.randn(2, nb, bs, bs, device=dev) * 0.1; b1 = torch.randn(2, nb, bs, device=dev) * 0.1
w2 = torch.randn(2, nb, bs, bs, device=dev) * 0.1; b2 = torch.randn(2, nb, bs, device=dev) * 0.1
save = torch.empty(lib.dlwp_afno2d_save_elems(B, H, W, Cc, nb, 1.0) * 2, device=dev)
g = [torch.zeros_like(t) for t in (w1, b1, w2, b2)]
def stamps():
    buf = (C.c_ulonglong * 32)(); lib.dlwp_debug_stamps_afno(buf); t = list(buf); return [t[i + 1] - t[i] for i in range(7)], t[7] - t[0]
for _ in range(3):
    assert lib.dlwp_afno2d_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), save.data_ptr(),
                               B, H, W, Cc, nb, 0.01, 1.0, None) == 0
    torch.cuda.synchronize()
print("fwd phases (cycles):", *stamps())
for _ in range(3):
    assert lib.dlwp_afno2d_bwd(gy.data_ptr(), save.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), gx.data_ptr(),
                               *[t.data_ptr() for t in g], B, H, W, Cc, nb, 0.01, 1.0, None) == 0
    torch.cuda.synchronize()
print("bwd phases (cycles):", *stamps())
