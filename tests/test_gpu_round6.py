"""Round-6 additions: the library's live per-kernel accounting (csrc/prof.hip, what bench.py's tertiary rooflines are built on), the
capture-safe zero fill of the tall column sum (advisor finding), and the dlwpbench SwinTransformer with BASELINE's window 7
against the CPU oracle (oracle/swin_ref.dlwp_swin with cfg["window_size"])."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_kernel_accounting_records_launches_with_their_algorithmic_work(cuda):
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm
    lib = L.load()
    M, N, K, T, C = 512, 256, 128, 1024, 96
    g = torch.Generator().manual_seed(0)
    A, B = torch.randn(M, K, generator=g).to(cuda), torch.randn(N, K, generator=g).to(cuda)
    Cm = torch.empty(M, N, device=cuda)
    x = torch.randn(T, C, generator=g).to(cuda)
    gam, bet = torch.ones(C, device=cuda), torch.zeros(C, device=cuda)
    y, mean, rstd = torch.empty_like(x), torch.empty(T, device=cuda), torch.empty(T, device=cuda)
    _gemm(A, B, Cm, M, N, K, K, K, N, 0, 1)          # not recorded: accounting is off
    with L.kernel_accounting() as acc:
        for _ in range(3):
            _gemm(A, B, Cm, M, N, K, K, K, N, 0, 1)
        L.check(lib.dlwp_layernorm_fwd(L.ptr(x), L.ptr(gam), L.ptr(bet), L.ptr(y), L.ptr(mean), L.ptr(rstd), T, C, 1e-5, L.stream()))
        # launches inside a capture are not recorded (an event pair in a graph measures nothing)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            _gemm(A, B, Cm, M, N, K, K, K, N, 0, 1)
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            _gemm(A, B, Cm, M, N, K, K, K, N, 0, 1)
    rows = {r["name"]: r for r in acc.rows}
    gemm = [r for n, r in rows.items() if n.startswith("gemm_kernel<")]
    assert len(gemm) == 1 and gemm[0]["calls"] == 4          # three + the side-stream one; the captured launch is absent
    assert gemm[0]["flops"] == pytest.approx(4 * 2.0 * M * N * K)
    assert gemm[0]["bytes"] == pytest.approx(4 * 4.0 * (M * K + N * K + M * N))
    assert 0 < gemm[0]["ms"] < 50
    ln = [r for n, r in rows.items() if n.startswith("layernorm_fwd")]
    assert len(ln) == 1 and ln[0]["calls"] == 1 and ln[0]["bytes"] == pytest.approx(T * C * 8.0 + 8.0 * T)
    assert acc.rows == sorted(acc.rows, key=lambda r: -r["ms"])
    with L.kernel_accounting() as acc2:          # a new scope starts empty
        pass
    assert acc2.rows == []
    assert torch.allclose(Cm, A @ B.T, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("T,N", [(32, 4096), (48, 1000)])
def test_tall_column_sum_overwrite_is_correct_under_graph_replay(cuda, T, N):
    """dlwp_colsum_ex(overwrite = 1) with more than 16 rows zero-fills its output before the atomic slab kernel: with a kernel, not
    hipMemsetAsync (captured memset nodes were seen writing garbage on later replays).  The SFNO position-embedding gradient takes
    this path at per-GPU batch > 16 inside the captured step."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(T)
    x = torch.randn(T, N, generator=g).to(cuda)
    out = torch.full((N,), 7.0, device=cuda)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        L.check(lib.dlwp_colsum_ex(L.ptr(x), L.ptr(out), T, N, 1, L.stream()))
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        L.check(lib.dlwp_colsum_ex(L.ptr(x), L.ptr(out), T, N, 1, L.stream()))
    for rep in range(4):
        out.fill_(float(rep) + 3.0)          # garbage the overwrite must not see
        x.copy_(torch.randn(T, N, generator=g))
        graph.replay()
        torch.cuda.synchronize()
        assert rel(out, x.double().sum(0)) <= 1e-5, rep


@pytest.mark.parametrize("H,W,B", [(32, 64, 2), (20, 36, 1), (28, 42, 1)])
def test_dlwp_swin_window7_matches_oracle(cuda, H, W, B):
    """dlwpbench SwinTransformer with the extra kwarg window_size = 7 (BASELINE configs[3]; the reference class fixes window = stage
    resolution and its block cannot pad) against oracle/swin_ref.dlwp_swin: nsbench BasicLayer arithmetic (pinned to
    the reference by c4_window7_golden.npz) with constant latitude / circular longitude padding.  fp32, output 1e-4, gradients 2e-3."""
    from dlwp_benchmark_amd import dlwpbench
    from oracle import swin_ref
    torch.manual_seed(5)
    cfg = dict(constant_channels=2, prescribed_channels=1, prognostic_channels=3, context_size=1, img_height=H, img_width=W,
               patch_size=1, embed_dim=16, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, window_size=7)
    m = dlwpbench.SwinTransformer(**cfg)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "relative_position_bias_table" in n:
                p.mul_(25.0)                       # the 0.02-scaled table would hide an index error
    g = torch.Generator().manual_seed(H)
    kw = dict(constants=torch.randn(B, 1, 2, H, W, generator=g), prescribed=torch.randn(B, 3, 1, H, W, generator=g),
              prognostic=torch.randn(B, 3, 3, H, W, generator=g))
    target = torch.randn(B, 2, 3, H, W, generator=g)
    p = {k: v.detach().clone() for k, v in m.state_dict().items()}
    names = {n for n, _ in m.named_parameters()}
    for k in names:
        p[k].requires_grad_(True)
    yr = swin_ref.dlwp_swin(kw["constants"], kw["prescribed"], kw["prognostic"], p, dict(cfg, patch_norm=True))
    torch.nn.functional.mse_loss(yr, target).backward()
    m = m.to(cuda).train()
    y = m(**{k: v.to(cuda) for k, v in kw.items()})
    assert rel(y, yr) <= 1e-4
    torch.nn.functional.mse_loss(y, target.to(cuda)).backward()
    for n, q in m.named_parameters():
        if q.grad is not None and p[n].grad is not None:
            assert rel(q.grad, p[n].grad) <= 2e-3, n


@pytest.mark.parametrize("M,N,K", [(8192, 384, 1536), (8192, 384, 384), (8192, 1152, 384), (16384, 192, 768), (65536 + 70, 96, 288),
                                   (4100, 288, 96), (1000, 576, 192)])
@pytest.mark.parametrize("form", ["nt_bias_res", "nt_bias_gelu_pre", "nn_plain", "nn_gelu_grad"])
def test_glds_gemm_96_wide_tiles_are_the_128_wide_ones_bit_for_bit(cuda, M, N, K, form):
    """gemm_glds_kernel<.., .., 96> (128 x 96 output tiles: csrc/token_ops.hip) against the 128 x 128 tiles of the same kernel: the K
    order of every output element is the same, so the results must be identical bits, for y = x W^T and gx = g W, both K-step depths,
    edge tiles in M, and the epilogues of the token layers.  The result itself is held to a float64 product."""
    import torch
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(cuda).to(BF)
    nt = form.startswith("nt")
    w = ((torch.randn(N, K, generator=g) if nt else torch.randn(K, N, generator=g)) / K ** 0.5).to(cuda).to(BF)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(cuda)
    zz = torch.randn(M, N, generator=g).to(cuda).to(BF)
    ref = x.double() @ (w.double().T if nt else w.double())
    outs = {}
    L.set_tuning("GEMM_GLDS_FORCE", 1)
    variants = [0, 2]
    try:
        for mode in variants:
            L.set_tuning("GEMM_GLDS_N96", mode)
            with L.gemm_precision("bf16"), L.kernel_accounting(shapes=True) as acc:
                if form == "nt_bias_res":
                    y = torch.empty(M, N, device=cuda)
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 0, None, res)
                    extra = None
                elif form == "nt_bias_gelu_pre":
                    y = torch.empty(M, N, device=cuda, dtype=BF)
                    extra = torch.empty(M, N, device=cuda, dtype=BF)
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 1, extra, None)
                elif form == "nn_plain":
                    y = torch.empty(M, N, device=cuda, dtype=BF)
                    _gemm(x, w, y, M, N, K, K, N, N, 0, 0)
                    extra = None
                else:
                    y = torch.empty(M, N, device=cuda, dtype=BF)
                    _gemm_batched(x, w, y, M, N, K, K, N, N, 0, 0, act=4, residual=zz)
                    extra = None
                torch.cuda.synchronize()
            names = [r["name"] for r in acc.rows]
            assert len(names) == 1 and names[0].startswith("gemm_glds_kernel"), names
            assert names[0].split(">")[0].endswith(", 96") == (mode == 2), names
            outs[mode] = (y, extra)
    finally:
        for k in ("GEMM_GLDS_N96", "GEMM_GLDS_FORCE"):
            L.set_tuning(k, None)
    first = outs[variants[0]]
    for v in variants[1:]:
        assert torch.equal(first[0], outs[v][0]), v
        if first[1] is not None:
            assert torch.equal(first[1], outs[v][1]), v
    y = outs[variants[-1]][0].double()
    if form == "nt_bias_res":
        want, tol = ref + bias.double() + res.double(), 2e-5
    elif form == "nt_bias_gelu_pre":
        want, tol = torch.nn.functional.gelu(ref + bias.double()), 1e-2
    elif form == "nn_plain":
        want, tol = ref, 1e-2
    else:
        zd = zz.double().requires_grad_()
        (gd,) = torch.autograd.grad(torch.nn.functional.gelu(zd).sum(), zd)
        want, tol = ref * gd, 1e-2
    assert ((y - want).abs().max() / want.abs().max()).item() <= tol


@pytest.mark.parametrize("T,C", [(65536, 96), (16384, 192), (32768, 192), (8192, 384), (16200, 768), (2049, 96), (9000, 200), (20000, 52), (2100, 256)])
def test_layernorm_backward_eight_wave_workgroups(cuda, T, C):
    """layernorm_bwd_vec_kernel runs the large inputs as 256 workgroups of eight waves (csrc/norm_ops.hip, ln_bwd_waves) and the widths
    96 / 192 / 384 / 768 take the three-chunk kernel (layernorm_bwd_vecn_kernel): gx and the gradients of gamma / beta against a float64
    reference for every variant, with the residual gradient and a bf16 upstream
    gradient as the C4 steps pass them, and row counts that are not multiples of the rows per workgroup."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(T, C, generator=g).to(cuda)
    gy = torch.randn(T, C, generator=g).to(cuda).bfloat16()
    ga = torch.randn(T, C, generator=g).to(cuda)
    gam = (1.0 + 0.1 * torch.randn(C, generator=g)).to(cuda)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    xh = (x.double() - mean.double()[:, None]) * rstd.double()[:, None]
    gyd = gy.double()
    want_g, want_b = (gyd * xh).sum(0), gyd.sum(0)
    gg_ = gyd * gam.double()
    want_x = rstd.double()[:, None] * (gg_ - gg_.mean(1, keepdim=True) - xh * (gg_ * xh).mean(1, keepdim=True)) + ga.double()
    tol = 2e-5 * T ** 0.5
    # (waves per workgroup of the one-chunk kernel, three-chunk kernel for the widths 96 / 192 / 384 / 768 on or off)
    for nw, v3 in ((4, 0), (8, 0), (8, 1)):
        L.set_tuning("LN_BWD_NW", nw)
        L.set_tuning("LN_BWD_V3", v3)
        try:
            gx = torch.empty_like(x)
            gg, gb = torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
            L.check(lib.dlwp_layernorm_bwd_ex(L.ptr(x), L.ptr(gam), L.ptr(mean), L.ptr(rstd), L.ptr(gy), 1, L.ptr(ga), L.ptr(gx), L.ptr(gg), L.ptr(gb),
                                              T, C, L.stream()))
            torch.cuda.synchronize()
        finally:
            L.set_tuning("LN_BWD_NW", None)
            L.set_tuning("LN_BWD_V3", None)
        assert (gx.double() - want_x).abs().max().item() <= 1e-5 * want_x.abs().max().item(), nw
        assert (gg.double() - want_g).abs().max().item() <= tol * want_g.abs().max().item(), nw
        assert (gb.double() - want_b).abs().max().item() <= tol * want_b.abs().max().item(), nw


@pytest.mark.parametrize("N,d,heads,B_,nW,masked", [(49, 24, 4, 1406, 703, True), (49, 48, 4, 380, 190, True), (49, 32, 6, 512, 64, False), (64, 16, 8, 300, 4, True)])
def test_window_attention_bf16_tensors_in_the_window_layout(cuda, N, d, heads, B_, nW, masked):
    """dlwp_window_attn_fwd_bf16 / _bwd_bf16 (qkv, out, gout, gqkv as bf16 arrays: the wave-per-window forward on raw bf16 fragments, the
    two-pass LDS-staged backward reading / writing bf16 rows) against the fp32-tensor entries on the same bf16-rounded values.
    Tolerance: the bf16 resolution of the outputs (4e-3 relative to the largest entry) plus the second rounding of the scaled q."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    BF = torch.bfloat16
    g = torch.Generator().manual_seed(N + d + heads)
    TB = 3 * N
    qkv = torch.randn(B_, N, 3 * heads * d, generator=g).to(cuda).to(BF)
    table = (0.5 * torch.randn(TB, heads, generator=g)).to(cuda)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(cuda) if masked else None
    gout = torch.randn(B_, N, heads * d, generator=g).to(cuda).to(BF)
    scale = d ** -0.5
    with L.gemm_precision("bf16"):
        assert lib.dlwp_window_attn_io_bf16_supported(N, d, TB, B_ * heads) == 1
        # fp32 tensors
        q32 = qkv.float()
        out32, lse32 = torch.empty(B_, N, heads * d, device=cuda), torch.empty(B_, heads, N, device=cuda)
        L.check(lib.dlwp_window_attn_fwd_qrange(L.ptr(q32), L.ptr(table), None, L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out32), L.ptr(lse32),
                                                B_, nW, N, TB, 1, heads, d, scale, 0, N, L.stream()))
        gq32, gt32 = torch.empty_like(q32), torch.zeros_like(table)
        dsum = torch.empty_like(lse32)
        slab = torch.empty(lib.dlwp_window_attn_bwd_slab_floats(B_, N, heads, TB), device=cuda)
        o32r = out32.to(BF).float()          # the bf16 path's backward reads the ROUNDED output
        L.check(lib.dlwp_window_attn_bwd_qrange(L.ptr(q32), L.ptr(table), None, L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(o32r), L.ptr(lse32),
                                                L.ptr(gout.float()), L.ptr(gq32), L.ptr(gt32), L.ptr(dsum), L.ptr(slab), B_, nW, N, TB, 1, heads, d,
                                                scale, 0, N, L.stream()))
        # bf16 tensors
        out16, lse16 = torch.empty(B_, N, heads * d, device=cuda, dtype=BF), torch.empty(B_, heads, N, device=cuda)
        L.check(lib.dlwp_window_attn_fwd_bf16(L.ptr(qkv), L.ptr(table), None, L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out16), L.ptr(lse16),
                                              B_, nW, N, TB, 1, heads, d, scale, L.stream()))
        gq16, gt16 = torch.empty_like(qkv), torch.zeros_like(table)
        L.check(lib.dlwp_window_attn_bwd_bf16(L.ptr(qkv), L.ptr(table), None, L.ptr(ia), L.ptr(ib), L.ptr(labels), L.ptr(out16), L.ptr(lse16),
                                              L.ptr(gout), L.ptr(gq16), L.ptr(gt16), B_, nW, N, TB, 1, heads, d, scale, L.stream()))
        torch.cuda.synchronize()
    rel = lambda a, b: ((a.double() - b.double()).abs().max() / b.double().abs().max()).item()   # noqa: E731
    assert rel(out16, out32) <= 1e-2
    assert rel(lse16, lse32) <= 2e-3
    assert rel(gq16, gq32) <= 2e-2
    assert rel(gt16, gt32) <= 2e-2


@pytest.mark.parametrize("T,C,B", [(8192, 384, 1), (65536, 96, 2), (16384, 192, 4), (16200, 768, 2), (9000, 200, 3), (3000, 50, 2)])
def test_layernorm_backward_second_output_is_the_scaled_bf16_gradient(cuda, T, C, B):
    """dlwp_layernorm_bwd_lowp: gx_bf16 == bf16(gx * row_scale[sample]) for every LayerNorm backward kernel (three-chunk, one-chunk, wide-row,
    scalar), gx itself and the column gradients unchanged against dlwp_layernorm_bwd_ex."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(T, C, generator=g).to(cuda)
    gy = torch.randn(T, C, generator=g).to(cuda).bfloat16()
    ga = torch.randn(T, C, generator=g).to(cuda)
    gam = (1.0 + 0.1 * torch.randn(C, generator=g)).to(cuda)
    scale = torch.tensor([0.0, 1.25, 1.0, 1.1][:B], device=cuda)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    gx0, gg0, gb0 = torch.empty_like(x), torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
    L.check(lib.dlwp_layernorm_bwd_ex(L.ptr(x), L.ptr(gam), L.ptr(mean), L.ptr(rstd), L.ptr(gy), 1, L.ptr(ga), L.ptr(gx0), L.ptr(gg0), L.ptr(gb0),
                                      T, C, L.stream()))
    gx1, gg1, gb1 = torch.empty_like(x), torch.zeros(C, device=cuda), torch.zeros(C, device=cuda)
    low = torch.empty(T, C, device=cuda, dtype=torch.bfloat16)
    L.check(lib.dlwp_layernorm_bwd_lowp(L.ptr(x), L.ptr(gam), L.ptr(mean), L.ptr(rstd), L.ptr(gy), 1, L.ptr(ga), L.ptr(gx1), L.ptr(gg1), L.ptr(gb1),
                                        T, C, L.ptr(low), L.ptr(scale), T // B, L.stream()))
    torch.cuda.synchronize()
    assert torch.equal(gx0, gx1)
    assert (gg0 - gg1).abs().max().item() <= 1e-4 * gg0.abs().max().item() and (gb0 - gb1).abs().max().item() <= 1e-4 * gb0.abs().max().item()
    want = (gx1.reshape(B, T // B, C) * scale[:, None, None]).reshape(T, C).bfloat16()
    assert torch.equal(low, want)


def test_transposed_weight_copies_and_the_input_gradient_products(cuda):
    """train_engine keeps [in][out] bf16 copies of the 2-D weights (dlwp_transpose_cast_bf16_many, one launch per step) and the Linear /
    Mlp backward computes gx = g W on them in the k-contiguous form: the copies are the transposes of the bf16 weights, bit for bit, and
    the gradients of a small token model agree with the [k][n] form (same operands, another summation order)."""
    import torch.nn as nn
    from dlwp_benchmark_amd import lib as L, token_ops as TO, train_engine as TE
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.norm = TO.LayerNorm(192)
            self.qkv = TO.Linear(192, 576)
            self.back = TO.Linear(576, 192)
            self.norm2 = TO.LayerNorm(192)
            self.mlp = TO.Mlp(192, 768)

        def forward(self, x):
            skip, t = TO.norm_fork(self.norm, x, gemm_input=True)
            x = self.back(self.qkv(t), residual=skip)
            skip, t = TO.norm_fork(self.norm2, x, gemm_input=True)
            return self.mlp(t, residual=skip)

    L.set_gemm_precision("bf16")
    L.set_storage("bf16")
    try:
        torch.manual_seed(3)
        net = Net().to(cuda).train()
        x = torch.randn(2, 4096, 192, device=cuda)
        target = torch.randn(2, 4096, 192, device=cuda)
        step = GraphedTrainStep(net, {"x": x}, target, lr=1e-3, use_graph=False)
        assert getattr(net, "_dlwp_flat16t", None) is not None
        grads = {}
        for nn_form in (True, False):
            TO.INPUT_GRAD_NN = nn_form
            step.grad.zero_()
            step._fwd_bwd()
            torch.cuda.synchronize()
            grads[nn_form] = step.grad.clone()
        for p in net.parameters():
            if p.dim() == 2:
                assert torch.equal(p._dlwp_bf16_t, p._dlwp_bf16.t().contiguous()), tuple(p.shape)
        a, b = grads[True].double(), grads[False].double()
        assert ((a - b).abs().max() / a.abs().max()).item() <= 2e-2
        assert ((a - b).norm() / a.norm()).item() <= 3e-3
    finally:
        TO.INPUT_GRAD_NN = False
        L.set_storage("fp32")
        L.set_gemm_precision("fp32")
