"""Whole-launch span of the forward `spatial` kernel (needs `make stamps`): earliest first instruction to latest last instruction
over all workgroups on the 100 MHz constant clock, next to the HIP-event time per launch of back-to-back launches."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'dlwp_benchmark_amd', 'libdlwpmi_stamps.so'))
V, I = C.c_void_p, C.c_int
lib.dlwp_fno_plan_create.argtypes = [I] * 5 + [C.POINTER(V)]
lib.dlwp_fno_spatial_fwd_probe.argtypes = [V] * 7 + [I, V]
lib.dlwp_debug_span_fno.argtypes = [V, I]
dev = 'cuda'
for B in (1, 4):
    Cc, H, W, m1, m2c = 32, 64, 64, 12, 7
    plan = V(); lib.dlwp_fno_plan_create(Cc, H, W, m1, m2c, C.byref(plan))
    x = torch.randn(B, Cc, H, W, device=dev); spec = torch.randn(B, m1, m2c, Cc, 2, device=dev) * 0.1
    wk = torch.randn(Cc, Cc, device=dev); bias = torch.zeros(Cc, device=dev); pre = torch.empty_like(x)
    x1 = torch.empty(B, H, m2c, Cc, 2, device=dev)
    def launch():
        assert lib.dlwp_fno_spatial_fwd_probe(plan, x.data_ptr(), spec.data_ptr(), wk.data_ptr(), bias.data_ptr(), pre.data_ptr(),
                                              x1.data_ptr(), B, None) == 0
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    spans = []
    buf = (C.c_ulonglong * 2)()
    lib.dlwp_debug_span_fno(buf, 1)
    for _ in range(20):
        launch(); torch.cuda.synchronize()
        lib.dlwp_debug_span_fno(buf, 1)
        spans.append((buf[1] - buf[0]) * 10)          # ns
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        launch()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B}: in-kernel span first-start -> last-end: median {sorted(spans)[10]} ns (min {min(spans)}, max {max(spans)}); "
          f"back-to-back launches: {e0.elapsed_time(e1) * 1e3 / 200:.2f} us each")
