"""Seeded shape fuzzing of the token GEMM (csrc/token_ops.hip: every Linear, patch embedding, head, spherical transform
and spectral-weight product of the AFNO / Swin / Pangu / SFNO models runs on it): ragged M / N / K, all four transpose
combinations, leading dimensions larger than the rows, every epilogue (bias, GELU / ReLU / soft-shrink, residual before or
after the activation, stored pre-activation, accumulate, GELU'-multiply), row sums, strided batches, split-K shapes; in
fp32 mode against torch fp64, in bf16-operand mode against the product of the bf16-rounded operands."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def _act(v, act, lam):
    if act == 1:
        return F.gelu(v)
    if act == 2:
        return F.relu(v)
    if act == 3:
        return F.softshrink(v, lam)
    return v


@pytest.mark.parametrize("seed", range(24))
@pytest.mark.parametrize("mode", ["fp32", "bf16"])
def test_gemm_random_shapes_and_epilogues(cuda, seed, mode):
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm_batched
    rng = np.random.default_rng(1000 + seed)
    M, N, K = (int(rng.integers(1, 200)) for _ in range(3))
    if seed % 6 == 0:
        K = int(rng.integers(1500, 5000))                      # long K, few tiles: the split-K path (no epilogue below)
    tA, tB = int(rng.integers(0, 2)), int(rng.integers(0, 2))
    nb1, nb2 = (1, 1) if seed % 3 else (int(rng.integers(1, 4)), int(rng.integers(1, 3)))
    pad = lambda n: n + int(rng.integers(0, 3)) * 4             # noqa: E731   leading dimensions >= the row length
    lda = pad(M if tA else K)
    ldb = pad(K if tB else N)
    ldc = pad(N)
    g = torch.Generator().manual_seed(2000 + seed)
    A = torch.randn(nb1, nb2, (K if tA else M), lda, generator=g, dtype=torch.float64)
    B = torch.randn(nb1, nb2, (N if tB else K), ldb, generator=g, dtype=torch.float64)
    opA = (A[..., :M].transpose(-1, -2) if tA else A[..., :K])           # [.., M, K]
    opB = (B[..., :K].transpose(-1, -2) if tB else B[..., :N])           # [.., K, N]
    split = seed % 6 == 0
    use_bias = (not split) and bool(rng.integers(0, 2))
    act = 0 if split else int(rng.choice([0, 1, 2, 3, 4, 7, 8]))          # 7: GELU storing GELU' as "preact"; 8: multiply by the residual
    use_res = (not split) and (act in (4, 8) or bool(rng.integers(0, 2)))
    res_pre = int(rng.integers(0, 2)) if (use_res and act not in (4, 8)) else 0
    want_pre = (not split) and (act == 7 or (act in (1, 2, 3) and bool(rng.integers(0, 2))))
    accumulate = int(rng.integers(0, 2))
    lam = 0.3
    bias = torch.randn(N, generator=g, dtype=torch.float64) if use_bias else None
    resid = torch.randn(nb1, nb2, M, ldc, generator=g, dtype=torch.float64) if use_res else None
    C0 = torch.randn(nb1, nb2, M, ldc, generator=g, dtype=torch.float64)
    rb = (lambda t: t.float().bfloat16().double()) if mode == "bf16" else (lambda t: t.float().double())
    prod = rb(opA) @ rb(opB)
    v = prod + (bias if use_bias else 0)
    if act == 4:
        z = resid[..., :N].float().double()
        zz = z.clone().requires_grad_(True)
        F.gelu(zz).sum().backward()
        want = v * zz.grad
        pre_want = None
    elif act == 8:
        want = v * resid[..., :N].float().double()
        pre_want = None
    elif act == 7:
        if use_res and res_pre:
            v = v + resid[..., :N].float().double()
        vv = v.clone().requires_grad_(True)
        want = F.gelu(vv)
        want.sum().backward()
        want, pre_want = want.detach(), vv.grad                 # the "pre-activation" output holds the derivative
        if use_res and not res_pre:
            want = want + resid[..., :N].float().double()
    else:
        if use_res and res_pre:
            v = v + resid[..., :N].float().double()
        pre_want = v
        want = _act(v, act, lam)
        if use_res and not res_pre:
            want = want + resid[..., :N].float().double()
    if accumulate:
        want = want + C0[..., :N].float().double()
    dev = lambda t: None if t is None else t.float().to(cuda).contiguous()      # noqa: E731
    Ad, Bd, Cd, bd, rd = dev(A), dev(B), dev(C0), dev(bias), dev(resid)
    pre = torch.zeros_like(Cd) if want_pre else None
    rows_a, rows_b = A.shape[2], B.shape[2]
    with L.gemm_precision(mode):
        _gemm_batched(Ad, Bd, Cd, M, N, K, lda, ldb, ldc, tA, tB, nb1, nb2, (nb2 * rows_a * lda, rows_a * lda),
                      (nb2 * rows_b * ldb, rows_b * ldb), (nb2 * M * ldc, M * ldc), bias=bd, act=act, act_param=lam, preact=pre,
                      residual=rd, sR=(nb2 * M * ldc, M * ldc), res_pre=res_pre, accumulate=accumulate)
    tol = 3e-5 if mode == "fp32" else 1e-4
    assert rel(Cd[..., :N], want) <= tol * max(1.0, (K / 64) ** 0.5), (M, N, K, tA, tB, nb1, nb2, act, use_bias, use_res, res_pre)
    if want_pre:
        assert rel(pre[..., :N], pre_want) <= tol * max(1.0, (K / 64) ** 0.5)
    # columns beyond N (the padding of ldc) are never written
    if ldc > N:
        assert torch.equal(Cd[..., N:], C0[..., N:].float().to(cuda))
