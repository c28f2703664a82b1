"""GPU parity of the one-launch token MLP for wide hidden layers (csrc/mlp_stream.hip: dlwp_mlp_stream_pack / _fwd / _bwd;
reference: Mlp.forward, /root/reference/src/nsbench/models/fourcastnet/fourcastnet.py:40-56) against a float64 restatement with
the kernel's rounding points (bf16 operands, hidden tensors stored as bf16) and against the two-GEMM autograd node it replaces
(token_ops._MlpFn under bf16 operands + bf16 storage).  Tolerances: stored bf16 tensors 2^-7 (max-norm), fp32 outputs 1e-2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rb(t):
    return t.float().to(BF).double()


def gelu64(z):
    return 0.5 * z * (1.0 + torch.erf(z / np.sqrt(2.0)))


def gelu_grad64(z):
    return 0.5 * (1.0 + torch.erf(z / np.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)


@pytest.mark.parametrize("T,Hd,x_bf16,res", [(16200, 3072, True, True), (100, 1024, False, False), (4099, 3072, False, True)])
def test_stream_kernels_match_float64_with_the_kernels_rounding(cuda, T, Hd, x_bf16, res):
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    E = 768
    assert lib.dlwp_mlp_stream_supported(E, Hd) == 1
    g = torch.Generator().manual_seed(41)
    x = torch.randn(T, E, generator=g)
    w1, w2 = torch.randn(Hd, E, generator=g) / E ** 0.5, torch.randn(E, Hd, generator=g) / Hd ** 0.5
    b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(E, generator=g)
    r = torch.randn(T, E, generator=g) if res else None
    gout = torch.randn(T, E, generator=g)
    w1d, w2d, b1d, b2d = (t.to(cuda) for t in (w1, w2, b1, b2))
    imgs = torch.empty(4, E * Hd, device=cuda, dtype=BF)
    L.check(lib.dlwp_mlp_stream_pack(L.ptr(w1d), L.ptr(w2d), E, Hd, L.ptr(imgs), L.stream()))
    xd = x.to(cuda).to(BF) if x_bf16 else x.to(cuda)
    x_lp = None if x_bf16 else torch.full((T, E), float("nan"), device=cuda, dtype=BF)
    rd = r.to(cuda) if res else None
    z, h = (torch.full((T, Hd), float("nan"), device=cuda, dtype=BF) for _ in range(2))
    y = torch.full((T, E), float("nan"), device=cuda)
    L.check(lib.dlwp_mlp_stream_fwd(L.ptr(xd), int(x_bf16), L.ptr(x_lp), L.ptr(imgs[0]), L.ptr(b1d), L.ptr(imgs[1]), L.ptr(b2d), L.ptr(rd),
                                    L.ptr(z), L.ptr(h), L.ptr(y), T, E, Hd, L.stream()))
    torch.cuda.synchronize()
    zr = rb(x) @ rb(w1).t() + b1.double()
    hr = rb(gelu64(zr))
    yr = hr @ rb(w2).t() + b2.double() + (r.double() if res else 0.0)
    if not x_bf16:
        assert torch.equal(x_lp.cpu(), x.to(BF))
    assert rel(z, zr) <= 2 ** -7 and rel(h, hr) <= 2 ** -7
    assert rel(y, yr) <= 1e-2
    # backward from the kernel's own stored pre-activation
    gd = gout.to(cuda)
    g_lp = torch.full((T, E), float("nan"), device=cuda, dtype=BF)
    gh = torch.full((T, Hd), float("nan"), device=cuda, dtype=BF)
    for gx_bf16 in (0, 1):
        gx = torch.full((T, E), float("nan"), device=cuda, dtype=BF if gx_bf16 else torch.float32)
        L.check(lib.dlwp_mlp_stream_bwd(L.ptr(gd), L.ptr(g_lp), L.ptr(imgs[2]), L.ptr(imgs[3]), L.ptr(z), L.ptr(gh), L.ptr(gx), gx_bf16, T, E, Hd,
                                        L.stream()))
        torch.cuda.synchronize()
        ghr = rb((rb(gout) @ rb(w2)) * gelu_grad64(z.cpu().double()))
        gxr = ghr @ rb(w1)
        assert torch.equal(g_lp.cpu(), gout.to(BF))
        assert rel(gh, ghr) <= 2 ** -7
        assert rel(gx, gxr) <= 1e-2
        assert torch.isfinite(gx.float()).all()


def test_stream_node_equals_the_two_gemm_node(cuda, monkeypatch):
    from dlwp_benchmark_amd import lib as L, token_ops
    monkeypatch.setattr(token_ops, "MLP_STREAM", True)          # opt-in path (DLWP_MLP_STREAM=1)
    T, E, Hd = 1000, 768, 3072
    g = torch.Generator().manual_seed(42)
    x = torch.randn(T, E, generator=g)
    w1, w2 = torch.randn(Hd, E, generator=g) / E ** 0.5, torch.randn(E, Hd, generator=g) / Hd ** 0.5
    b1, b2 = 0.1 * torch.randn(Hd, generator=g), 0.1 * torch.randn(E, generator=g)
    r, gout = torch.randn(T, E, generator=g), torch.randn(T, E, generator=g)
    res = {}
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        try:
            for name, fn in (("stream", token_ops._MlpStreamFn), ("gemm", token_ops._MlpFn)):
                leaves = [t.clone().to(cuda).requires_grad_(True) for t in (x, w1, b1, w2, b2, r)]
                if name == "stream":
                    assert fn.applies(leaves[0], leaves[1], leaves[3])
                out = fn.apply(*leaves)
                out.backward(gout.to(cuda))
                res[name] = [out] + [p.grad for p in leaves]
        finally:
            L.SHADOW_ACTIVE = False
            L.set_storage("fp32")
    for i, (a, b) in enumerate(zip(res["stream"], res["gemm"])):
        assert rel(a, b) <= 1e-2, (i, rel(a, b))
