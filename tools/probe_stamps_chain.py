#!/usr/bin/env python3
"""Phase stamps (workgroup 0, thread 0; needs `make stamps`) and stand-alone timing of the one-launch SFNO block tail
(csrc/mlp_chain.hip) at the C3 shape: T = 8192 tokens (B = 4), C = 256, hidden = 512."""
import ctypes as C
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from dlwp_benchmark_amd.token_ops import _TailBwdArgs, _TailFwdArgs      # noqa: E402

stamps = "--nostamps" not in sys.argv
lib = C.CDLL(os.path.join(here, "..", "dlwp_benchmark_amd", "libdlwpmi_stamps.so" if stamps else "libdlwpmi.so"))
V, I = C.c_void_p, C.c_int
lib.dlwp_mlp_chain_pack.argtypes = [V, I, I, I, V, V]
lib.dlwp_sfno_tail_fwd.argtypes = [V, V]
lib.dlwp_sfno_tail_bwd.argtypes = [V, V]
if stamps:
    lib.dlwp_debug_stamps_chain.argtypes = [V]
dev = "cuda"
BF = torch.bfloat16
for T in [int(a) for a in sys.argv[1:] if a.isdigit()] or [8192, 32768]:
    Cc, Hd = 256, 512
    x, y, g = (torch.randn(T, Cc, device=dev) for _ in range(3))
    ws, w1, w2 = torch.randn(Cc, Cc, device=dev) / 16, torch.randn(Hd, Cc, device=dev) / 16, torch.randn(Cc, Hd, device=dev) / 22
    bs, b1, b2 = torch.randn(Cc, device=dev), torch.randn(Hd, device=dev), torch.randn(Cc, device=dev)
    imgs = torch.empty(6, Cc * Hd, device=dev, dtype=BF)
    for i, (w, r, c, tr) in enumerate(((ws, Cc, Cc, 0), (w1, Hd, Cc, 0), (w2, Cc, Hd, 0), (w2, Hd, Cc, 1), (w1, Cc, Hd, 1), (ws, Cc, Cc, 1))):
        assert lib.dlwp_mlp_chain_pack(w.data_ptr(), r, c, tr, imgs[i].data_ptr(), None) == 0
    e = lambda n, dt=BF: torch.empty(T, n, device=dev, dtype=dt)
    x_lp, z0, t, z1, h, out = e(Cc), e(Cc), e(Cc), e(Hd), e(Hd), e(Cc, torch.float32)
    g_lp, gh, gt, gt_lp, gx = e(Cc), e(Hd), e(Cc, torch.float32), e(Cc), e(Cc, torch.float32)
    fa = _TailFwdArgs(x.data_ptr(), y.data_ptr(), imgs[0].data_ptr(), imgs[1].data_ptr(), imgs[2].data_ptr(), bs.data_ptr(), b1.data_ptr(),
                      b2.data_ptr(), x_lp.data_ptr(), z0.data_ptr(), t.data_ptr(), z1.data_ptr(), h.data_ptr(), out.data_ptr(), T, Cc, Hd, 1)
    ba = _TailBwdArgs(g.data_ptr(), imgs[3].data_ptr(), imgs[4].data_ptr(), imgs[5].data_ptr(), z1.data_ptr(), z0.data_ptr(), g_lp.data_ptr(),
                      gh.data_ptr(), gt.data_ptr(), gt_lp.data_ptr(), gx.data_ptr(), T, Cc, Hd, 1)
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    names = ["issue+x wait+commit", "barrier0", "stage1 mma", "epi1", "barrier1", "stage2 mma", "epi2", "barrier2", "stage3 mma", "epi3 issue",
             "store drain"]
    for tag, fn, arg in (("fwd", lib.dlwp_sfno_tail_fwd, fa), ("bwd", lib.dlwp_sfno_tail_bwd, ba)):
        for cold in (0, 1):
            ts = []
            for _ in range(20):
                if cold:
                    flush.zero_()
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                assert fn(C.byref(arg), None) == 0
                ev1.record()
                torch.cuda.synchronize()
                ts.append(ev0.elapsed_time(ev1) * 1e3)
            ts.sort()
            line = f"T={T} {tag} {'cold' if cold else 'warm'}: median {ts[len(ts) // 2]:.1f} us (min {ts[0]:.1f})"
            if stamps:
                buf = (C.c_ulonglong * 32)()
                lib.dlwp_debug_stamps_chain(buf)
                s = list(buf)
                line += "  cycles: " + ", ".join(f"{n} {s[i + 1] - s[i]}" for i, n in enumerate(names)) + f"; total {s[11] - s[0]}"
            print(line, flush=True)
    # back-to-back chain of 50 launches in a graph-like stream order (no host gaps matter: events around the whole chain)
    for tag, fn, arg in (("fwd", lib.dlwp_sfno_tail_fwd, fa), ("bwd", lib.dlwp_sfno_tail_bwd, ba)):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(50):
            fn(C.byref(arg), None)
        ev1.record()
        torch.cuda.synchronize()
        print(f"T={T} {tag} x50 back to back: {ev0.elapsed_time(ev1) * 1e3 / 50:.1f} us per launch", flush=True)
