"""GPU parity of the one-launch SFNO block tail (csrc/mlp_chain.hip: dlwp_mlp_chain_pack, dlwp_sfno_tail_fwd / _bwd) against
(i) a float64 restatement with the kernel's rounding points (operands and stored hidden tensors rounded to bf16, fp32-or-better
sums; block: torch_harmonics' SFNO block, constructed at /root/reference/src/dlwpbench/models/fno/fno.py:183-200, SURVEY.md
App. A-2) and (ii) the three-GEMM autograd node it replaces (token_ops._SkipMlpFn under bf16 storage).
Tolerances: stored bf16 tensors within one bf16 ulp-scale step (2^-7 relative, max-norm), fp32 outputs 1e-2 relative max-norm
(a hidden value that lands on the other side of a bf16 rounding boundary moves the output by ~2^-9 of one term)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def rb(t):
    """round to bf16, back to float64"""
    return t.float().to(BF).double()


def gelu64(z):
    return 0.5 * z * (1.0 + torch.erf(z / np.sqrt(2.0)))


def gelu_grad64(z):
    return 0.5 * (1.0 + torch.erf(z / np.sqrt(2.0))) + z * torch.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi)


@pytest.mark.parametrize("rows,cols,transpose", [(64, 64, 0), (128, 64, 1), (256, 512, 0), (512, 256, 1), (48, 96, 1)])
def test_pack_image_is_the_fragment_order_of_the_matrix(cuda, rows, cols, transpose):
    from dlwp_benchmark_amd import lib as L
    g = torch.Generator().manual_seed(1)
    W = torch.randn((cols, rows) if transpose else (rows, cols), generator=g)
    img = torch.empty(rows * cols, device=cuda, dtype=BF)
    Wd = W.to(cuda)
    L.check(L.load().dlwp_mlp_chain_pack(L.ptr(Wd), rows, cols, transpose, L.ptr(img), L.stream()))
    Wp = (W.t() if transpose else W).to(BF)                       # the matrix the image describes, [rows][cols]
    KS = cols // 32
    want = Wp.reshape(rows // 16, 16, KS, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)      # [tile][kk][g][r][e]
    assert torch.equal(img.cpu(), want)


def _tail_inputs(T, C, Hd, seed):
    g = torch.Generator().manual_seed(seed)
    x, y, gout = (torch.randn(T, C, generator=g) for _ in range(3))
    ws = torch.randn(C, C, generator=g) / np.sqrt(C)
    w1 = torch.randn(Hd, C, generator=g) / np.sqrt(C)
    w2 = torch.randn(C, Hd, generator=g) / np.sqrt(Hd)
    bs, b1, b2 = (0.1 * torch.randn(n, generator=g) for n in (C, Hd, C))
    return x, y, gout, ws, bs, w1, b1, w2, b2


def _images(L, dev, ws, w1, w2):
    C, Hd = ws.shape[0], w1.shape[0]
    imgs = torch.empty(6, C * Hd, device=dev, dtype=BF)
    plan = ((ws, C, C, 0), (w1, Hd, C, 0), (w2, C, Hd, 0), (w2, Hd, C, 1), (w1, C, Hd, 1), (ws, C, C, 1))
    held = []
    for i, (w, rows, cols, tr) in enumerate(plan):
        held.append(w.to(dev).contiguous())
        L.check(L.load().dlwp_mlp_chain_pack(L.ptr(held[-1]), rows, cols, tr, L.ptr(imgs[i]), L.stream()))
    torch.cuda.synchronize()
    return imgs


@pytest.mark.parametrize("T,C,Hd,outer", [(8192, 256, 512, 1), (1000, 256, 512, 0), (77, 64, 128, 1), (4096, 128, 256, 1),
                                           (33, 256, 512, 1), (16421, 256, 512, 1)])
def test_tail_forward_and_backward_match_float64_with_the_kernels_rounding(cuda, T, C, Hd, outer):
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _TailBwdArgs, _TailFwdArgs
    import ctypes
    lib = L.load()
    assert lib.dlwp_mlp_chain_supported(C, Hd) == 1
    x, y, gout, ws, bs, w1, b1, w2, b2 = _tail_inputs(T, C, Hd, 11)
    d = lambda t: t.to(cuda).contiguous()
    imgs = _images(L, cuda, ws, w1, w2)
    xd, yd, gd, bsd, b1d, b2d = d(x), d(y), d(gout), d(bs), d(b1), d(b2)
    e = lambda n, dt=BF: torch.full((T, n), float("nan"), device=cuda, dtype=dt)
    x_lp, z0, t, z1, h, out = e(C), e(C), e(C), e(Hd), e(Hd), e(C, torch.float32)
    a = _TailFwdArgs(L.ptr(xd), L.ptr(yd), L.ptr(imgs[0]), L.ptr(imgs[1]), L.ptr(imgs[2]), L.ptr(bsd), L.ptr(b1d), L.ptr(b2d),
                     L.ptr(x_lp), L.ptr(z0), L.ptr(t), L.ptr(z1), L.ptr(h), L.ptr(out), T, C, Hd, outer)
    L.check(lib.dlwp_sfno_tail_fwd(ctypes.byref(a), L.stream()))
    torch.cuda.synchronize()
    # float64 with the same rounding points
    X, Y = x.double(), y.double()
    z0r = Y + rb(x) @ rb(ws).t() + bs.double()
    tr = rb(gelu64(z0r))
    z1r = tr @ rb(w1).t() + b1.double()
    hr = rb(gelu64(z1r))
    outr = hr @ rb(w2).t() + b2.double() + (X if outer else 0.0)
    assert torch.equal(x_lp.cpu(), x.to(BF))
    # (the arrays called z0 / z1 hold GELU'(z0) / GELU'(z1): what the backward kernel multiplies by)
    assert rel(z0, gelu_grad64(z0r)) <= 2 ** -7 and rel(t, tr) <= 2 ** -7
    assert rel(z1, gelu_grad64(z1r)) <= 2 ** -6 and rel(h, hr) <= 2 ** -6
    assert rel(out, outr) <= 1e-2
    # backward, from the kernel's own stored activation derivatives (so that the comparison isolates the backward arithmetic)
    g_lp, gh, gt, gt_lp, gx = e(C), e(Hd), e(C, torch.float32), e(C), e(C, torch.float32)
    b = _TailBwdArgs(L.ptr(gd), L.ptr(imgs[3]), L.ptr(imgs[4]), L.ptr(imgs[5]), L.ptr(z1), L.ptr(z0), L.ptr(g_lp), L.ptr(gh),
                     L.ptr(gt), L.ptr(gt_lp), L.ptr(gx), T, C, Hd, outer)
    L.check(lib.dlwp_sfno_tail_bwd(ctypes.byref(b), L.stream()))
    torch.cuda.synchronize()
    G = gout.double()
    ghr = rb((rb(gout) @ rb(w2)) * z1.cpu().double())
    gtr = (ghr @ rb(w1)) * z0.cpu().double()
    gxr = rb(gtr) @ rb(ws) + (G if outer else 0.0)
    assert torch.equal(g_lp.cpu(), gout.to(BF))
    assert rel(gh, ghr) <= 2 ** -7
    assert rel(gt, gtr) <= 1e-2 and rel(gt_lp, gtr) <= 1e-2
    assert rel(gx, gxr) <= 1e-2
    for buf in (out, gt, gx, z0, t, z1, h, gh, gt_lp):
        assert torch.isfinite(buf.float()).all()


@pytest.mark.parametrize("T,C,Hd", [(2048, 256, 512), (200, 64, 128)])
def test_chain_node_equals_the_three_gemm_node(cuda, T, C, Hd):
    """autograd: outputs, both input gradients and all six parameter gradients of token_ops._SkipMlpChainFn against
    token_ops._SkipMlpFn (the three-GEMM tail) under bf16 operands + bf16 storage."""
    from dlwp_benchmark_amd import lib as L, token_ops
    x, y, gout, ws, bs, w1, b1, w2, b2 = _tail_inputs(T, C, Hd, 12)
    res = {}
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        try:
            for name, fn in (("chain", token_ops._SkipMlpChainFn), ("gemm", token_ops._SkipMlpFn)):
                leaves = [t.clone().to(cuda).requires_grad_(True) for t in (y, x, ws.reshape(C, C, 1, 1), bs, w1.reshape(Hd, C, 1, 1), b1,
                                                                           w2.reshape(C, Hd, 1, 1), b2)]
                if name == "chain":
                    assert fn.applies(leaves[1], leaves[2], leaves[4], leaves[6])
                out = fn.apply(*leaves, True)
                out.backward(gout.to(cuda))
                res[name] = [out] + [p.grad for p in leaves]
        finally:
            L.SHADOW_ACTIVE = False
            L.set_storage("fp32")
    for i, (a, b) in enumerate(zip(res["chain"], res["gemm"])):
        assert rel(a, b) <= 1e-2, (i, rel(a, b))


def test_tail_pack_equals_six_single_packs(cuda):
    from dlwp_benchmark_amd import lib as L
    C, Hd = 128, 256
    _, _, _, ws, _, w1, _, w2, _ = _tail_inputs(16, C, Hd, 3)
    want = _images(L, cuda, ws, w1, w2)
    got = torch.zeros(6, C * Hd, device=cuda, dtype=BF)
    wsd, w1d, w2d = ws.to(cuda), w1.to(cuda), w2.to(cuda)
    L.check(L.load().dlwp_sfno_tail_pack(L.ptr(wsd), L.ptr(w1d), L.ptr(w2d), C, Hd, L.ptr(got), L.stream()))
    for i, n in enumerate((C * C, Hd * C, C * Hd, Hd * C, C * Hd, C * C)):
        assert torch.equal(got[i, :n].cpu(), want[i, :n].cpu()), i


@pytest.mark.parametrize("T,nseg,dims", [(8192, 4, ((256, 512), (512, 256), (256, 256))), (1000, 3, ((64, 128), (128, 64))),
                                          (333, 1, ((256, 512),)), (4100, 8, ((136, 72), (8, 8), (512, 256), (256, 256)))])
def test_segmented_weight_gradients_match_float64(cuda, T, nseg, dims):
    """dlwp_wgrad_segments: gW += sum_s g_s^T x_s and gb += sum_s colsum(g_s) over bf16 operand segments, accumulated into
    non-zero gradient buffers; ragged token counts (partial K-steps, partial slices) and ragged widths (tiles past the edge);
    two runs agree bit for bit in the weight gradients (ordered slab reduction)."""
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(21)
    layers, ref = [], []
    for (N, K) in dims:
        gs = [torch.randn(T, N, generator=g).to(BF) for _ in range(nseg)]
        xs = [torch.randn(T, K, generator=g).to(BF) for _ in range(nseg)]
        gw0, gb0 = torch.randn(N, K, generator=g), torch.randn(N, generator=g)
        ref.append((gw0.double() + sum(a.double().t() @ b.double() for a, b in zip(gs, xs)),
                    gb0.double() + sum(a.double().sum(0) for a in gs)))
        layers.append(([t.to(cuda) for t in gs], [t.to(cuda) for t in xs], gw0, gb0, N, K))
    runs = []
    for _ in range(2):
        args = [(gs, xs, gw0.clone().to(cuda), gb0.clone().to(cuda), True, (N, K)) for gs, xs, gw0, gb0, N, K in layers]
        outs = token_ops._weight_grad_segments(args)
        assert all(o == (None, None) for o in outs)
        torch.cuda.synchronize()
        runs.append([(a[2].cpu(), a[3].cpu()) for a in args])
    for (gw, gb), (gwr, gbr) in zip(runs[0], ref):
        assert rel(gw, gwr) <= 2e-5, rel(gw, gwr)
        assert rel(gb, gbr) <= 2e-5, rel(gb, gbr)
    for (gw_a, _), (gw_b, _) in zip(*runs):
        assert torch.equal(gw_a, gw_b)


def test_rollout_scope_defers_the_tail_weight_gradients_to_the_last_lead_time(cuda):
    """Three applications of one block tail inside a spectral_weight_scope: the weight gradients arrive in ONE segmented product
    (launched by the last backward pass through the weights) and equal the sum of the per-application gradients of the
    three-GEMM node."""
    from dlwp_benchmark_amd import lib as L, sht, token_ops
    T, C, Hd = 512, 64, 128
    x, y, gout, ws, bs, w1, b1, w2, b2 = _tail_inputs(T, C, Hd, 13)
    calls = []
    orig = token_ops._weight_grad_segments

    def spy(layers):
        calls.append(len(layers[0][0]))
        return orig(layers)
    res = {}
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        token_ops._weight_grad_segments = spy
        try:
            for name, fn in (("chain", token_ops._SkipMlpChainFn), ("gemm", token_ops._SkipMlpFn)):
                P = [t.clone().to(cuda).requires_grad_(True) for t in (ws.reshape(C, C, 1, 1), bs, w1.reshape(Hd, C, 1, 1), b1,
                                                                      w2.reshape(C, Hd, 1, 1), b2)]
                xin = x.clone().to(cuda).requires_grad_(True)
                with sht.spectral_weight_scope():
                    t = xin
                    for _ in range(3):                       # the block output feeds the next application (a rollout's lead times)
                        t = fn.apply(y.to(cuda), t, *P, True)
                    t.backward(gout.to(cuda))
                res[name] = [t, xin.grad] + [p.grad for p in P]
        finally:
            token_ops._weight_grad_segments = orig
            L.SHADOW_ACTIVE = False
            L.set_storage("fp32")
    assert calls == [3]
    for i, (a, b) in enumerate(zip(res["chain"], res["gemm"])):
        assert rel(a, b) <= 2e-2, (i, rel(a, b))
