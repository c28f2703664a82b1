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


@pytest.mark.parametrize("T,C,eps", [(1024, 64, 1e-6), (333, 40, 1e-5), (50, 384, 1e-5), (1001, 768, 1e-5), (77, 1000, 1e-6), (130, 260, 1e-5),
                                     (4099, 96, 1e-5), (2050, 192, 1e-6), (8192, 384, 1e-5), (2051, 768, 1e-5)])     # the last four: the three-chunk kernels
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
    from dlwp_benchmark_amd import lib as L
    L.set_tuning("LN_FWD_V3", 2)          # the three-chunk forward from 2048 rows on (by default from 4 M elements)
    try:
        y = ln(xd)
    finally:
        L.set_tuning("LN_FWD_V3", None)
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


@pytest.mark.parametrize("T,C,Hd,N,res,slots", [(1024, 64, 256, 64, True, True), (300, 96, 192, 96, True, False),
                                                  (77, 13, 29, 13, False, False), (512, 128, 256, 40, False, True)])
def test_mlp_single_node(cuda, T, C, Hd, N, res, slots):
    """token_ops.mlp: fc2(GELU(fc1 x)) (+ residual) with the GELU derivative applied in the epilogue of the g W2 product
    (GEMM act 4); checked against torch's fp32 autograd, with and without preallocated gradient slots."""
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(T, C, generator=g)
    w1, b1 = torch.randn(Hd, C, generator=g) / C ** 0.5, torch.randn(Hd, generator=g) * 0.1
    w2, b2 = torch.randn(N, Hd, generator=g) / Hd ** 0.5, torch.randn(N, generator=g) * 0.1
    r = torch.randn(T, N, generator=g) if res else None
    gy = torch.randn(T, N, generator=g)
    ref = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)] + ([r.clone().requires_grad_(True)] if res else [])
    y_ref = F.linear(F.gelu(F.linear(ref[0], ref[1], ref[2])), ref[3], ref[4])
    if res:
        y_ref = y_ref + ref[5]
    y_ref.backward(gy)
    dev = [t.to(cuda).requires_grad_(True) for t in (x, w1, b1, w2, b2)] + ([r.to(cuda).requires_grad_(True)] if res else [])
    params = [torch.nn.Parameter(t.detach()) for t in dev[1:5]]
    if slots:                                   # fused accumulation straight into the gradient buffers, on top of a value
        for p in params:
            p.grad = torch.full_like(p, 0.25)
    y = token_ops.mlp(dev[0], *params, dev[5] if res else None)
    y.backward(gy.to(cuda))
    assert rel(y, y_ref) <= 1e-4
    assert rel(dev[0].grad, ref[0].grad) <= 5e-4
    for p, q in zip(params, ref[1:5]):
        assert rel(p.grad - (0.25 if slots else 0.0), q.grad) <= 5e-4
    if res:
        assert rel(dev[5].grad, ref[5].grad) <= 1e-6


@pytest.mark.parametrize("T,C,Hd,outer", [(1024, 64, 128, True), (260, 48, 96, False)])
def test_skip_mlp_single_node(cuda, T, C, Hd, outer):
    """token_ops.skip_mlp (tail of an SFNO block): out = fc2(GELU(fc1 t)) (+ x), t = GELU(y + skip(x))."""
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(4)
    mk = lambda *s: torch.randn(*s, generator=g)      # noqa: E731
    y, x = mk(T, C), mk(T, C)
    ws, bs = mk(C, C) / C ** 0.5, mk(C) * 0.1
    w1, b1 = mk(Hd, C) / C ** 0.5, mk(Hd) * 0.1
    w2, b2 = mk(C, Hd) / Hd ** 0.5, mk(C) * 0.1
    gout = mk(T, C)
    ref = [t.clone().requires_grad_(True) for t in (y, x, ws, bs, w1, b1, w2, b2)]
    t_ref = F.gelu(ref[0] + F.linear(ref[1], ref[2], ref[3]))
    o_ref = F.linear(F.gelu(F.linear(t_ref, ref[4], ref[5])), ref[6], ref[7])
    if outer:
        o_ref = o_ref + ref[1]
    o_ref.backward(gout)
    dev = [t.to(cuda).requires_grad_(True) for t in (y, x, ws, bs, w1, b1, w2, b2)]
    out = token_ops.skip_mlp(*dev, outer)
    out.backward(gout.to(cuda))
    assert rel(out, o_ref) <= 1e-4
    for got, want in zip(dev, ref):
        assert rel(got.grad, want.grad) <= 5e-4


def test_gemm_bf16_deep_k(cuda):
    """bf16-operand mode with K not a multiple of the 64-deep K-step and transposed operands (weight-gradient product,
    split-K path included): equals the fp32 product of the bf16-rounded operands."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm
    g = torch.Generator().manual_seed(9)
    rb = lambda t: t.bfloat16().float()    # noqa: E731
    for (M, N, K) in [(128, 192, 4000), (200, 72, 100), (64, 64, 36)]:
        A = torch.randn(K, M, generator=g)          # op(A) = A^T [M, K]
        B = torch.randn(K, N, generator=g)
        C_ = torch.empty(M, N, device=cuda)
        with L.gemm_precision("bf16"):
            _gemm(A.to(cuda), B.to(cuda), C_, M, N, K, M, N, N, 1, 0)
        assert rel(C_, rb(A).t() @ rb(B)) <= 2e-5


@pytest.mark.parametrize("B,H,W,Cin,O,k,act", [(2, 8, 6, 16, 8, 2, 1), (1, 5, 7, 12, 10, 2, 0), (3, 4, 4, 8, 5, 1, 1), (2, 3, 5, 6, 4, 3, 1)])
def test_upconv_tokens_matches_conv_transpose(cuda, B, H, W, Cin, O, k, act):
    """UpConvT2d.forward_tokens (GEMM in the weight's layout + dlwp_upconv_shuffle) against nn.ConvTranspose2d (+ GELU) on NCHW."""
    from dlwp_benchmark_amd import token_ops
    g = torch.Generator().manual_seed(B * 100 + H * 10 + k)
    ref = torch.nn.ConvTranspose2d(Cin, O, kernel_size=k, stride=k)
    with torch.no_grad():
        ref.weight.copy_(torch.randn(ref.weight.shape, generator=g) * 0.3)
        ref.bias.copy_(torch.randn(O, generator=g))
    up = token_ops.UpConvT2d(Cin, O, kernel_size=k, stride=k)
    up.load_state_dict(ref.state_dict())
    up = up.to(cuda)
    x = torch.randn(B, H, W, Cin, generator=g)
    gy = torch.randn(B, H * k, W * k, O, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = ref(xr.permute(0, 3, 1, 2))
    if act:
        yr = F.gelu(yr)
    yr = yr.permute(0, 2, 3, 1)
    yr.backward(gy)
    xd = x.to(cuda).requires_grad_(True)
    y = up.forward_tokens(xd, act=act)
    y.backward(gy.to(cuda))
    assert rel(y, yr) <= 1e-4
    assert rel(xd.grad, xr.grad) <= 5e-4
    assert rel(up.weight.grad, ref.weight.grad) <= 5e-4
    assert rel(up.bias.grad, ref.bias.grad) <= 5e-4


@pytest.mark.parametrize("T,N", [(3, 1000003), (4, 4096), (16, 64), (17, 64), (1000, 768), (16200, 3072), (33, 7), (257, 1028)])
def test_colsum_accumulates_the_column_sums(cuda, T, N):
    """dlwp_colsum: out[n] += sum_t g[t][n] -- the flat form (T <= 16: batch sum behind a position embedding's gradient, no
    atomics) and the tall form (row slabs, 16-byte loads when aligned, float atomics), aligned and odd widths."""
    from dlwp_benchmark_amd import lib as L
    g = torch.Generator().manual_seed(T + N)
    x = torch.randn(T, N, generator=g).to(cuda)
    out = torch.randn(N, generator=g).to(cuda)
    want = out.double() + x.double().sum(0)
    L.check(L.load().dlwp_colsum(L.ptr(x), L.ptr(out), T, N, L.stream()))
    torch.cuda.synchronize()
    assert ((out.double() - want).abs().max() / want.abs().max()).item() <= 2e-6


def test_weight_gradients_of_three_layers_in_one_launch(cuda):
    """dlwp_weight_grad_group: gW_i (+)= g_i^T x_i and the bias gradients of three layers in one launch (fp32 and bf16 operands
    mixed, accumulation into existing gradients and a fresh output) against float64."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _weight_grad_group
    gen = torch.Generator().manual_seed(11)
    T = 2048
    shapes = [(256, 512), (512, 256), (256, 256)]           # (N out, K in)
    items, want = [], []
    with L.gemm_precision("bf16"):
        for i, (N, K) in enumerate(shapes):
            g = torch.randn(T, N, generator=gen).to(cuda)
            x = torch.randn(T, K, generator=gen).to(cuda)
            if i == 0:
                g = g.to(torch.bfloat16)
            if i != 2:
                x = x.to(torch.bfloat16)
            wslot = torch.randn(N, K, generator=gen).to(cuda) if i < 2 else None
            bslot = torch.randn(N, generator=gen).to(cuda) if i == 0 else None
            has_bias = i != 1
            # the matrix units round fp32 operands to bf16: the reference product uses the rounded values
            gr, xr = g.to(torch.bfloat16).double(), x.to(torch.bfloat16).double()
            want.append(((wslot.double() if wslot is not None else 0) + gr.T @ xr,
                         ((bslot.double() if bslot is not None else 0) + gr.sum(0)) if has_bias else None))       # (row sums of the rounded operand)
            items.append((g, x, wslot, bslot, has_bias, (N, K)))
        outs = _weight_grad_group(items)
    torch.cuda.synchronize()
    for (g, x, wslot, bslot, has_bias, _), (gw, gb), (ww, wb) in zip(items, outs, want):
        got_w = wslot if wslot is not None else gw
        assert (gw is None) == (wslot is not None)
        assert ((got_w.double() - ww).abs().max() / ww.abs().max()).item() <= 2e-5
        if has_bias:
            got_b = bslot if bslot is not None else gb
            assert ((got_b.double() - wb).abs().max() / wb.abs().max()).item() <= 2e-5
        else:
            assert gb is None


def test_add2d_many_and_colsum_overwrite(cuda):
    """dlwp_add2d_many: dst[r][c] += src[r * rs + c * cs] for several matrices in one launch (padded, column-offset and transposed
    sources); dlwp_colsum_ex(overwrite = 1) writes the column sums where dlwp_colsum adds them.  Exact in fp32 (one add each)."""
    import ctypes as C
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.sfno_ops import _Add2dDesc
    lib = L.load()
    g = torch.Generator().manual_seed(3)
    E, cin, KP, HW = 24, 10, 32, 35
    d0 = torch.randn(E, cin, generator=g).to(cuda)
    s0 = torch.randn(E, KP, generator=g).to(cuda)                  # padded rows: the first cin columns count
    d1 = torch.randn(E, E + 6, generator=g).to(cuda)               # two sources side by side in one destination
    s1a, s1b = torch.randn(E, E, generator=g).to(cuda), torch.randn(E, KP, generator=g).to(cuda)
    d2 = torch.randn(E, HW, generator=g).to(cuda)
    s2 = torch.randn(HW, E, generator=g).to(cuda)                  # transposed source
    want = [d0 + s0[:, :cin], torch.cat([d1[:, :E] + s1a, d1[:, E:] + s1b[:, :6]], dim=1), d2 + s2.t()]
    descs = (_Add2dDesc * 8)()
    descs[0] = _Add2dDesc(L.ptr(d0), L.ptr(s0), cin, KP, 1, E, cin)
    descs[1] = _Add2dDesc(L.ptr(d1), L.ptr(s1a), E + 6, E, 1, E, E)
    descs[2] = _Add2dDesc(L.ptr(d1) + 4 * E, L.ptr(s1b), E + 6, KP, 1, E, 6)
    descs[3] = _Add2dDesc(L.ptr(d2), L.ptr(s2), HW, 1, E, E, HW)
    L.check(lib.dlwp_add2d_many(C.cast(descs, C.c_void_p), 4, L.stream()))
    for got, w in zip((d0, d1, d2), want):
        assert torch.equal(got, w)
    assert lib.dlwp_add2d_many(C.cast(descs, C.c_void_p), 9, L.stream()) != 0
    x = torch.randn(5, 1000, generator=g).to(cuda)
    out = torch.full((1000,), 7.0, device=cuda)
    L.check(lib.dlwp_colsum_ex(L.ptr(x), L.ptr(out), 5, 1000, 1, L.stream()))
    ref = x[0] + x[1] + x[2] + x[3] + x[4]
    assert torch.allclose(out, ref, atol=1e-6)
    L.check(lib.dlwp_colsum_ex(L.ptr(x), L.ptr(out), 5, 1000, 0, L.stream()))
    assert torch.allclose(out, 2 * ref, atol=1e-5)


def test_colsum_of_a_bf16_array(cuda):
    from dlwp_benchmark_amd import lib as L
    x = torch.randn(300, 1536, device=cuda).to(torch.bfloat16)
    out = torch.ones(1536, device=cuda)
    L.check(L.load().dlwp_colsum_bf16(L.ptr(x), L.ptr(out), 300, 1536, L.stream()))
    ref = 1.0 + x.double().sum(0)
    assert ((out.double() - ref).abs().max() / ref.abs().max()).item() < 1e-5
    assert L.load().dlwp_colsum_bf16(L.ptr(x), L.ptr(out), 8, 1536, L.stream()) != 0
