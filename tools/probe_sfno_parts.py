#!/usr/bin/env python3
"""Time the pieces of one SFNO block at the C3 shapes (32x64, C=256): SHT, spectral weights, inverse SHT, skip, MLP.
    python tools/probe_sfno_parts.py [fp32|bf16] [B=16] [fused|gemm]"""
import sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import torch
from dlwp_benchmark_amd import sht, lib as L
from dlwp_benchmark_amd.token_ops import Conv1x1

dev = torch.device("cuda:0")
B, H, W, C = (int(sys.argv[2]) if len(sys.argv) > 2 else 16), 32, 64, 256
FUSED = (sys.argv[3] != "gemm") if len(sys.argv) > 3 else True
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
L.set_gemm_precision(prec)
fwd = sht.RealSHT(H, W, 32, 32, "equiangular", fused=FUSED).to(dev)
inv = sht.InverseRealSHT(H, W, 32, 32, "equiangular", fused=FUSED).to(dev)
x = torch.randn(B, H, W, C, device=dev, requires_grad=True)
w = (torch.randn(C, C, 32, 2, device=dev) * 0.05).requires_grad_(True)
skip = Conv1x1(C, C).to(dev)
fc1, fc2 = Conv1x1(C, 2 * C).to(dev), Conv1x1(2 * C, C).to(dev)


def timeit(name, f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    print(f"{prec} {name:28s} {e0.elapsed_time(e1) / n * 1e3:9.1f} us")


X = fwd(x).detach().requires_grad_(True)
Y = sht.dhconv(X, w).detach().requires_grad_(True)
y = inv(Y).detach()
gX, gx = torch.randn_like(X), torch.randn_like(x)


def fb(fn, inp, g):
    def run():
        inp.grad = None
        out = fn(inp)
        out.backward(g)
    return run


timeit("SHT fwd", lambda: fwd(x))
timeit("SHT fwd+bwd", fb(fwd, x, gX))
timeit("dhconv fwd", lambda: sht.dhconv(X, w))
timeit("dhconv fwd+bwd", fb(lambda t: sht.dhconv(t, w), X, gX))
timeit("iSHT fwd", lambda: inv(Y))
timeit("iSHT fwd+bwd", fb(inv, Y, gx))
timeit("skip GEMM fwd (+GELU)", lambda: skip(x, act=1, residual=y, res_pre=True))
timeit("skip fwd+bwd", fb(lambda t: skip(t, act=1, residual=y, res_pre=True), x, gx))
timeit("MLP fwd", lambda: fc2(fc1(x, act=1), residual=x))
timeit("MLP fwd+bwd", fb(lambda t: fc2(fc1(t, act=1), residual=t), x, gx))
