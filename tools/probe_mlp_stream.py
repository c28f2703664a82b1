#!/usr/bin/env python3
"""Token MLP at the FourCastNet-scale shape (16200 tokens, 768 -> 3072 -> 768): the one-launch streamed kernels
(csrc/mlp_stream.hip) against the two-GEMM node, forward and backward (activation products only and with the weight gradients)."""
import os
import sys

import torch

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, ".."))
from dlwp_benchmark_amd import lib as L, token_ops      # noqa: E402

dev = "cuda"
BF = torch.bfloat16


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for T in [int(a) for a in sys.argv[1:]] or [16200]:
    E, Hd = 768, 3072
    g = torch.Generator().manual_seed(1)
    w1 = (torch.randn(Hd, E, generator=g) / E ** 0.5).to(dev).requires_grad_(True)
    w2 = (torch.randn(E, Hd, generator=g) / Hd ** 0.5).to(dev).requires_grad_(True)
    b1, b2 = torch.zeros(Hd, device=dev, requires_grad=True), torch.zeros(E, device=dev, requires_grad=True)
    x = torch.randn(T, E, generator=g).to(dev).to(BF).requires_grad_(True)
    r = torch.randn(T, E, generator=g).to(dev)
    gout = torch.randn(T, E, generator=g).to(dev)
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        for p in (w1, w2, b1, b2):
            p.grad = torch.zeros_like(p)
        lib = L.load()
        imgs = torch.empty(4, E * Hd, device=dev, dtype=BF)
        t_pack = timed(lambda: L.check(lib.dlwp_mlp_stream_pack(L.ptr(w1.detach()), L.ptr(w2.detach()), E, Hd, L.ptr(imgs), L.stream())))
        z, h = torch.empty(T, Hd, device=dev, dtype=BF), torch.empty(T, Hd, device=dev, dtype=BF)
        y = torch.empty(T, E, device=dev)
        xd = x.detach()
        t_f = timed(lambda: L.check(lib.dlwp_mlp_stream_fwd(L.ptr(xd), 1, None, L.ptr(imgs[0]), L.ptr(b1.detach()), L.ptr(imgs[1]), L.ptr(b2.detach()),
                                                            L.ptr(r), L.ptr(z), L.ptr(h), L.ptr(y), T, E, Hd, L.stream())))
        g_lp, gh, gx = torch.empty(T, E, device=dev, dtype=BF), torch.empty(T, Hd, device=dev, dtype=BF), torch.empty(T, E, device=dev, dtype=BF)
        t_b = timed(lambda: L.check(lib.dlwp_mlp_stream_bwd(L.ptr(gout), L.ptr(g_lp), L.ptr(imgs[2]), L.ptr(imgs[3]), L.ptr(z), L.ptr(gh), L.ptr(gx), 1,
                                                            T, E, Hd, L.stream())))
        flops = 4.0 * T * E * Hd
        print(f"T={T}: stream pack {t_pack:.1f} us, fwd {t_f:.1f} us ({flops / t_f / 1e6:.0f} TFLOP/s), bwd {t_b:.1f} us ({flops / t_b / 1e6:.0f} TFLOP/s)",
              flush=True)
        for name, fn in (("stream", token_ops._MlpStreamFn), ("gemm", token_ops._MlpFn)):
            def fwd():
                return fn.apply(x, w1, b1, w2, b2, r)

            def fwd_bwd():
                fn.apply(x, w1, b1, w2, b2, r).backward(gout)
            with torch.no_grad():
                tf = timed(fwd)
            tfb = timed(fwd_bwd)
            print(f"T={T}: {name:6s} node forward {tf:.1f} us, forward + backward (with weight gradients) {tfb:.1f} us", flush=True)
        L.SHADOW_ACTIVE = False
        L.set_storage("fp32")
