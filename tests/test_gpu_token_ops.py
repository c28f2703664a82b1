"""GPU parity of the token-level kernels (MFMA GEMM with fused epilogue, LayerNorm, token MLP) against
torch's CPU fp32 reference of the same ops (floating-point kernels: the oracle here is plain torch)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("T,K,N,act,res", [(1024, 64, 256, 1, False), (1024, 256, 64, 0, True), (77, 13, 29, 1, True),
                                            (300, 96, 288, 0, False)])
def test_linear_fwd_bwd(cuda, T, K, N, act, res):
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(T, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    r = torch.randn(T, N, generator=g) if res else None
    gy = torch.randn(T, N, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w, b)] + ([r.clone().requires_grad_(True)] if res else [])
    y_ref = F.linear(ref_in[0], ref_in[1], ref_in[2])
    if act:
        y_ref = F.gelu(y_ref)
    if res:
        y_ref = y_ref + ref_in[3]
    y_ref.backward(gy)
    dev_in = [t.to(cuda).requires_grad_(True) for t in (x, w, b)] + ([r.to(cuda).requires_grad_(True)] if res else [])
    y = token_ops.linear(dev_in[0], dev_in[1], dev_in[2], act, dev_in[3] if res else None)
    y.backward(gy.to(cuda))
    assert rel(y, y_ref) <= 1e-4
    for got, ref in zip(dev_in, ref_in):
        assert rel(got.grad, ref.grad) <= 5e-4


@pytest.mark.parametrize("T,C,eps", [(1024, 64, 1e-6), (333, 40, 1e-5), (50, 384, 1e-5)])
def test_layernorm_fwd_bwd(cuda, T, C, eps):
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(2)
    x = torch.randn(T, C, generator=g) * 2 + 0.5
    ln_ref = torch.nn.LayerNorm(C, eps=eps)
    with torch.no_grad():
        ln_ref.weight.copy_(torch.randn(C, generator=g))
        ln_ref.bias.copy_(torch.randn(C, generator=g))
    ln = token_ops.LayerNorm(C, eps=eps)
    ln.load_state_dict(ln_ref.state_dict())
    ln = ln.to(cuda)
    gy = torch.randn(T, C, generator=g)
    xr = x.clone().requires_grad_(True)
    ln_ref(xr).backward(gy)
    xd = x.to(cuda).requires_grad_(True)
    y = ln(xd)
    y.backward(gy.to(cuda))
    assert rel(y, ln_ref(x)) <= 1e-4
    assert rel(xd.grad, xr.grad) <= 5e-4
    assert rel(ln.weight.grad, ln_ref.weight.grad) <= 5e-4
    assert rel(ln.bias.grad, ln_ref.bias.grad) <= 5e-4


@pytest.mark.parametrize("T,K,N", [(4096, 40, 120), (8192, 192, 768), (515, 52, 36), (257, 13, 29)])
def test_linear_fused_grad_accumulation(cuda, T, K, N):
    """Parameters with a preallocated .grad (train_engine.flatten_parameters) receive their gradients straight from
    the kernels (split-K weight-gradient GEMM with the bias gradient as a by-product), accumulated over two uses."""
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(5)
    lin_ref = torch.nn.Linear(K, N)
    lin = token_ops.Linear(K, N)
    lin.load_state_dict(lin_ref.state_dict())
    lin = lin.to(cuda)
    x1, x2 = torch.randn(T, K, generator=g), torch.randn(T, K, generator=g)
    gy = torch.randn(T, N, generator=g)
    (lin_ref(x1) + 0.5 * lin_ref(x2)).backward(gy)
    for p in lin.parameters():
        p.grad = torch.zeros_like(p)
    slots = [p.grad.data_ptr() for p in lin.parameters()]
    (lin(x1.to(cuda)) + 0.5 * lin(x2.to(cuda))).backward(gy.to(cuda))
    assert [p.grad.data_ptr() for p in lin.parameters()] == slots
    assert rel(lin.weight.grad, lin_ref.weight.grad) <= 5e-4
    assert rel(lin.bias.grad, lin_ref.bias.grad) <= 5e-4


@pytest.mark.parametrize("T,C", [(4096, 40), (2048, 768), (100, 1536)])
def test_layernorm_fused_grad_accumulation(cuda, T, C):
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(6)
    x = torch.randn(T, C, generator=g)
    gy = torch.randn(T, C, generator=g)
    ln_ref = torch.nn.LayerNorm(C)
    with torch.no_grad():
        ln_ref.weight.copy_(torch.randn(C, generator=g))
    ln = token_ops.LayerNorm(C)
    ln.load_state_dict(ln_ref.state_dict())
    ln = ln.to(cuda)
    xr = x.clone().requires_grad_(True)
    ln_ref(xr).backward(gy)
    for p in ln.parameters():
        p.grad = torch.zeros_like(p)
    xd = x.to(cuda).requires_grad_(True)
    ln(xd).backward(gy.to(cuda))
    assert rel(xd.grad, xr.grad) <= 5e-4
    assert rel(ln.weight.grad, ln_ref.weight.grad) <= 5e-4
    assert rel(ln.bias.grad, ln_ref.bias.grad) <= 5e-4


@pytest.mark.parametrize("T,K,N", [(1024, 256, 512), (300, 96, 288), (77, 13, 29)])
def test_linear_bf16_operand_mode(cuda, T, K, N):
    """dlwp_set_gemm_precision(bf16): operands rounded to bf16, fp32 accumulation -- equals an fp32 product of the
    bf16-rounded operands up to summation order."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(8)
    x = torch.randn(T, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) * 0.1
    gy = torch.randn(T, N, generator=g)
    rb = lambda t: t.bfloat16().float()    # noqa: E731
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    y_ref = F.linear(xr, wr, b)
    xd, wd, bd = (t.to(cuda).requires_grad_(True) for t in (x, w, b))
    with L.gemm_precision("bf16"):
        y = token_ops.linear(xd, wd, bd)
        y.backward(gy.to(cuda))
    assert L.load().dlwp_get_gemm_precision() == 0
    assert rel(y, y_ref) <= 1e-5
    # backward products round gy / x / w to bf16 as well
    gx_ref = rb(gy) @ rb(w)
    gw_ref = rb(gy).t() @ rb(x)
    assert rel(xd.grad, gx_ref) <= 1e-5
    assert rel(wd.grad, gw_ref) <= 1e-4
    assert rel(bd.grad, rb(gy).sum(0)) <= 1e-4
