"""bf16 STORAGE for the token GEMMs (dlwp_gemm_mixed, dlwp_cast_bf16, lib.set_storage): a bf16 array in memory must give
exactly what the bf16-operand GEMM gives on the same values held in fp32 (the matrix units round the operands to bf16 either
way), outputs are the fp32 results rounded once, and a train step with bf16 hidden activations + the engine's bf16 weight copy
stays within bf16 rounding of the fp32-storage step."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


def _call(lib, L, A, B, C, M, N, K, tA, tB, bias, act, preact, residual, dt):
    lda = A.shape[1]
    ldb = B.shape[1]
    L.check(lib.dlwp_gemm_mixed(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, lda, ldb, N, tA, tB, L.ptr(bias), act, L.ptr(preact),
                                L.ptr(residual), 0, None, dt, L.stream()))


@pytest.mark.parametrize("M,N,K", [(256, 192, 128), (100, 72, 52), (77, 29, 13), (64, 256, 1000)])
@pytest.mark.parametrize("tA,tB", [(0, 1), (0, 0), (1, 0), (1, 1)])
def test_gemm_mixed_equals_fp32_storage_of_rounded_values(cuda, M, N, K, tA, tB):
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(M + N + K + 2 * tA + tB)
    A = torch.randn((K, M) if tA else (M, K), generator=g).to(cuda)
    B = torch.randn((N, K) if tB else (K, N), generator=g).to(cuda)
    bias = torch.randn(N, generator=g).to(cuda)
    res = torch.randn(M, N, generator=g).to(cuda)
    A16, B16, R16 = A.to(BF), B.to(BF), res.to(BF)
    with L.gemm_precision("bf16"):
        # reference: fp32 arrays holding the bf16-rounded values, fp32 outputs
        C0, Z0 = torch.empty(M, N, device=cuda), torch.empty(M, N, device=cuda)
        _call(lib, L, A16.float(), B16.float(), C0, M, N, K, tA, tB, bias, 1, Z0, R16.float(), 0)
        for dt in (1, 2, 3, 4, 8, 15, 7, 11):
            a = A16 if dt & 1 else A16.float()
            b = B16 if dt & 2 else B16.float()
            r = R16 if dt & 8 else R16.float()
            C = torch.empty(M, N, device=cuda, dtype=BF if dt & 4 else torch.float32)
            Z = torch.empty_like(C)
            _call(lib, L, a, b, C, M, N, K, tA, tB, bias, 1, Z, r, dt)
            if dt & 4:
                assert torch.equal(C, C0.to(BF)) and torch.equal(Z, Z0.to(BF)), dt
            else:
                assert torch.equal(C, C0) and torch.equal(Z, Z0), dt
        # act 4 (multiply by GELU'(saved pre-activation)) with a bf16 pre-activation
        G0 = torch.empty(M, N, device=cuda)
        Af, Bf, Rf = A16.float(), B16.float(), R16.float()          # kept alive across the launch
        L.check(lib.dlwp_gemm_batched_mixed(L.ptr(Af), L.ptr(Bf), L.ptr(G0), M, N, K, A.shape[1], B.shape[1], N,
                                            tA, tB, 1, 1, 0, 0, 0, 0, 0, 0, None, 0, 0, 4, 0.0, None, L.ptr(Rf), 0, 0,
                                            0, 0, 0, L.stream()))
        G = torch.empty(M, N, device=cuda, dtype=BF)
        L.check(lib.dlwp_gemm_batched_mixed(L.ptr(A16), L.ptr(B16), L.ptr(G), M, N, K, A.shape[1], B.shape[1], N, tA, tB, 1, 1,
                                            0, 0, 0, 0, 0, 0, None, 0, 0, 4, 0.0, None, L.ptr(R16), 0, 0, 0, 0, 15, L.stream()))
        assert torch.equal(G, G0.to(BF))


def test_gemm_mixed_batched_strides_are_elements(cuda):
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(5)
    nb, M, N, K = 3, 40, 24, 36
    A = torch.randn(nb, M, K, generator=g).to(cuda).to(BF)
    B = torch.randn(nb, N, K, generator=g).to(cuda).to(BF)
    C = torch.empty(nb, M, N, device=cuda, dtype=BF)
    with L.gemm_precision("bf16"):
        L.check(lib.dlwp_gemm_batched_mixed(L.ptr(A), L.ptr(B), L.ptr(C), M, N, K, K, K, N, 0, 1, nb, 1, M * K, 0, N * K, 0,
                                            M * N, 0, None, 0, 0, 0, 0.0, None, None, 0, 0, 0, 0, 7, L.stream()))
    ref = torch.einsum("bmk,bnk->bmn", A.double(), B.double())
    assert ((C.double() - ref).abs().max() / ref.abs().max()).item() < 1e-2      # one bf16 rounding of the output


def test_gemm_mixed_refuses_accumulating_into_bf16(cuda):
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    A = torch.zeros(8, 8, device=cuda)
    C = torch.zeros(8, 8, device=cuda, dtype=BF)
    rc = lib.dlwp_gemm_mixed(L.ptr(A), L.ptr(A), L.ptr(C), 8, 8, 8, 8, 8, 8, 0, 1, None, 0, None, None, 1, None, 4, L.stream())
    assert rc != 0 and b"fp32 output" in lib.dlwp_last_error()


@pytest.mark.parametrize("n", [1, 7, 4096, 100003])
def test_cast_bf16_rounds_like_torch(cuda, n):
    from dlwp_benchmark_amd import lib as L
    g = torch.Generator().manual_seed(n)
    x = (torch.randn(n, generator=g) * 10 ** torch.randint(-6, 6, (n,), generator=g).float()).to(cuda)
    y = torch.empty(n, device=cuda, dtype=BF)
    L.check(L.load().dlwp_cast_bf16(L.ptr(x), L.ptr(y), n, L.stream()))
    assert torch.equal(y, x.to(BF))


class _Toy(torch.nn.Module):
    """LayerNorm -> token MLP (+ residual) -> linear head: the building blocks whose storage changes."""

    def __init__(self, E=64, Hd=256, out=8):
        super().__init__()
        from dlwp_benchmark_amd.token_ops import LayerNorm, Linear, Mlp
        self.norm, self.mlp, self.head = LayerNorm(E), Mlp(E, Hd), Linear(E, out)

    def forward(self, x):
        skip, t = self.norm.fork(x)
        return self.head(self.mlp(t, residual=skip))


def test_train_step_with_bf16_storage_tracks_fp32_storage(cuda):
    """Same bf16-operand arithmetic, hidden activations / weight copies stored as bf16: the losses of a few Adam steps agree
    with the fp32-storage run to bf16 rounding, and the stored tensors really are bf16."""
    from dlwp_benchmark_amd import lib as L, token_ops
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    g = torch.Generator().manual_seed(4)
    x = torch.randn(4, 300, 64, generator=g).to(cuda)
    y = torch.randn(4, 300, 8, generator=g).to(cuda)
    losses = {}
    seen = {"fp32": set(), "bf16": set()}
    orig, orig_group = token_ops._gemm, token_ops._weight_grad_group
    mode = "fp32"

    def spy(A, B, C, *a, **kw):
        seen[mode].add((A.dtype, B.dtype, C.dtype))
        return orig(A, B, C, *a, **kw)

    def spy_group(items):          # the weight gradients of the MLP's two layers go out as one grouped launch
        for g2, x2, *_ in items:
            seen[mode].add((g2.dtype, x2.dtype, torch.float32))
        return orig_group(items)
    with L.gemm_precision("bf16"):
        for mode in ("fp32", "bf16"):
            L.set_storage(mode)
            try:
                torch.manual_seed(0)
                m = _Toy().to(cuda)
                token_ops._gemm, token_ops._weight_grad_group = spy, spy_group
                step = GraphedTrainStep(m, {"x": x}, y, lr=1e-2)
                losses[mode] = [step().item() for _ in range(6)]
                if mode == "bf16":
                    w = m.mlp.fc1.weight
                    assert w._dlwp_bf16.dtype == BF and w._dlwp_bf16.shape == w.shape and w._dlwp_bf16.abs().sum().item() > 0
                    # the copy is refreshed at the top of a step: it holds the weights the LAST step started from
                    assert ((w._dlwp_bf16.float() - w.detach()).abs().max() < 0.05).item()
            finally:
                token_ops._gemm, token_ops._weight_grad_group = orig, orig_group
                L.set_storage("fp32")
    f32 = torch.float32
    assert seen["fp32"] == {(f32, f32, f32)}
    # fc1: x (fp32) . W1 (bf16 copy) -> h (bf16);  fc2: h . W2 -> y (fp32);  gx: gh (bf16) . W1;  gW1: gh^T x;  gW2: g^T h
    assert {(f32, BF, BF), (BF, BF, f32), (BF, f32, f32), (f32, BF, f32)} <= seen["bf16"]
    a, b = torch.tensor(losses["fp32"]), torch.tensor(losses["bf16"])
    assert a[-1] < a[0]                       # it trains
    assert ((a - b).abs() / a).max().item() < 2e-2, (losses["fp32"], losses["bf16"])


class _ToyBlock(nn.Module):
    """A transformer block's GEMM skeleton (qkv -> a stand-in for attention -> proj + skip -> fc1 / fc2 + skip) with the four weight
    gradients handed to one token_ops.WgradBatch per application, as the Pangu blocks do."""

    def __init__(self, C, Hd, drop):
        super().__init__()
        from dlwp_benchmark_amd import token_ops as TO
        self.norm1, self.norm2 = TO.LayerNorm(C), TO.LayerNorm(C)
        self.qkv, self.proj, self.mlp = TO.Linear(C, 3 * C), TO.Linear(C, C), TO.Mlp(C, Hd)
        self.drop_path = TO.DropPath(drop)

    def forward(self, x):
        from dlwp_benchmark_amd import token_ops as TO
        C = x.shape[-1]
        wb = TO.WgradBatch()
        skip, t = TO.norm_fork(self.norm1, x, gemm_input=True)
        q = self.qkv(t, wbatch=wb)
        t = q[..., :C] * torch.tanh(q[..., C:2 * C]) + q[..., 2 * C:]
        skip, t = TO.norm_fork(self.norm2, self.drop_path.branch(self.proj, t, skip, wbatch=wb), gemm_input=True)
        return self.drop_path.branch(self.mlp, t, skip, wbatch=wb)


@pytest.mark.parametrize("B,T,C,Hd,drop,applications", [(2, 1024, 192, 768, 0.3, 1), (1, 520, 96, 384, 0.0, 1), (3, 256, 64, 128, 0.2, 2)])
def test_block_weight_gradients_in_one_launch_equal_the_per_layer_products(cuda, B, T, C, Hd, drop, applications):
    """token_ops.WgradBatch (dlwp_wgrad_segments over the block's four (g, x) pairs, bf16 storage, gradient slots) against the
    per-layer split-K products on the same bf16 operands: every parameter gradient within 2e-3 of the max norm (fp32 sums in a
    different order; where the batch forces a bf16 copy of a gradient the per-layer path reads as fp32, one more bf16 rounding of
    that operand: 1e-2), input gradient likewise."""
    from dlwp_benchmark_amd import lib as L, token_ops as TO
    from dlwp_benchmark_amd.train_engine import flatten_parameters, refresh_bf16_weights
    g = torch.Generator().manual_seed(C + T)
    x0 = torch.randn(B, T, C, generator=g).to(cuda)
    gy = torch.randn(B, T, C, generator=g).to(cuda)
    masks = [(torch.rand(B, generator=g) > drop).float().to(cuda) / (1 - drop) for _ in range(2 * applications)]
    res, launches = {}, {}
    orig = TO._weight_grad_segments
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        try:
            for on in (True, False):
                torch.manual_seed(1)
                m = _ToyBlock(C, Hd, drop).to(cuda).train()
                flat, grad = flatten_parameters(m)
                refresh_bf16_weights(m)
                it = iter(masks)
                m.drop_path.mask = lambda batch, device: next(it)
                count = []
                TO._weight_grad_segments = lambda layers: (count.append(len(layers)), orig(layers))[1]
                TO.WGRAD_BATCH = on
                L.SHADOW_ACTIVE = True
                try:
                    x = x0.clone().requires_grad_(True)
                    y = x
                    for _ in range(applications):
                        y = m(y)
                    y.backward(gy)
                finally:
                    L.SHADOW_ACTIVE = False
                    TO.WGRAD_BATCH = True
                    TO._weight_grad_segments = orig
                res[on] = [y.detach(), x.grad] + [p.grad.clone() for p in m.parameters()]
                launches[on] = count
        finally:
            L.set_storage("fp32")
    assert launches[True] == [4] * applications and launches[False] == []
    for a, b in zip(res[True], res[False]):
        assert (a - b).abs().max().item() <= 1e-2 * b.abs().max().item() + 1e-12
    assert all(p.abs().sum().item() > 0 for p in res[True][2:])


def test_bf16_storage_needs_bf16_operands(cuda):
    from dlwp_benchmark_amd import lib as L
    with pytest.raises(L.DlwpError):
        L.set_storage("bf16")


def test_sfno_train_step_with_bf16_storage_tracks_fp32_storage(cuda):
    """C3-shaped SFNO (smaller width): with bf16 storage the transform chain keeps its GEMM-to-GEMM intermediates (longitude
    spectra, Legendre spectra, filtered spectra) and the expanded spectral weights as bf16; the loss trajectory must follow the
    fp32-storage run (same bf16-operand arithmetic) to bf16 rounding, and the chain really runs on bf16 arrays."""
    from dlwp_benchmark_amd import dlwpbench, lib as L, sht, token_ops
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    g = torch.Generator().manual_seed(8)
    B, T = 2, 3
    kw = dict(constants=torch.randn(B, 1, 4, 32, 64, generator=g).to(cuda), prescribed=torch.randn(B, T, 1, 32, 64, generator=g).to(cuda),
              prognostic=torch.randn(B, T, 5, 32, 64, generator=g).to(cuda))
    target = torch.randn(B, T - 1, 5, 32, 64, generator=g).to(cuda)
    seen = {"fp32": set(), "bf16": set()}
    orig = sht._gemm_batched
    mode = "fp32"

    def spy(A, Bm, C, *a, **k):
        seen[mode].add((A.dtype, Bm.dtype, C.dtype))
        return orig(A, Bm, C, *a, **k)
    losses = {}
    with L.gemm_precision("bf16"):
        for mode in ("fp32", "bf16"):
            L.set_storage(mode)
            try:
                torch.manual_seed(3)
                m = dlwpbench.SFNO2DModule(constant_channels=4, prescribed_channels=1, prognostic_channels=5, grid="equiangular",
                                           num_layers=2, scale_factor=1, embed_dim=64, context_size=1, height=32, width=64,
                                           big_skip=True, pos_embed=True, use_mlp=True, normalization_layer="none").to(cuda)
                sht._gemm_batched = spy
                step = GraphedTrainStep(m, kw, target, lr=2e-3)
                losses[mode] = [step().item() for _ in range(5)]
            finally:
                sht._gemm_batched = orig
                L.set_storage("fp32")
    f32 = torch.float32
    assert seen["fp32"] == {(f32, f32, f32)}
    # the per-degree weight products run on bf16 arrays; the transforms themselves are the one-launch bf16 kernels (csrc/sht_bf16.hip)
    assert (BF, BF, BF) in seen["bf16"]
    a, b = torch.tensor(losses["fp32"]), torch.tensor(losses["bf16"])
    assert a[-1] < a[0]
    assert ((a - b).abs() / a).max().item() < 3e-2, (losses["fp32"], losses["bf16"])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 384, 192), (8192, 576, 192), (130, 128, 64), (4100, 768, 3072), (32768, 384, 96)])
@pytest.mark.parametrize("epi", ["plain", "bias_gelu_preact_bf16", "residual_fp32", "gelu_grad_mul", "res_pre_accumulate"])
def test_lds_dma_bf16_gemm_matches_the_register_staged_kernel(cuda, monkeypatch, M, N, K, epi):
    """y = x W^T with both operands bf16 arrays: the 128 x 128 x 64 LDS-DMA kernel (csrc/token_ops.hip, gemm_glds_nt_kernel)
    against the register-staged kernel (DLWP_GEMM_NOGLDS is read once per process, so the oracle here is a float64 product of
    the bf16-rounded operands) for every epilogue the token layers use, with edge tiles in M and N."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(cuda).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda)
    ref = x.double() @ w.double().T
    with L.gemm_precision("bf16"):
        if epi == "plain":
            y = torch.empty(M, N, device=cuda)
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1)
            want = ref
        elif epi == "bias_gelu_preact_bf16":
            y = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
            z = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 1, z, None)
            pre = ref + bias.double()
            assert ((z.double() - pre).abs().max() / pre.abs().max()).item() <= 1e-2
            want = torch.nn.functional.gelu(pre)
        elif epi == "residual_fp32":
            r = torch.randn(M, N, generator=g).to(cuda)
            y = torch.empty(M, N, device=cuda)
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 0, None, r)
            want = ref + bias.double() + r.double()
        elif epi == "gelu_grad_mul":
            zz = torch.randn(M, N, generator=g).to(cuda).to(torch.bfloat16)
            y = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
            _gemm_batched(x, w, y, M, N, K, K, K, N, 0, 1, act=4, residual=zz)
            zd = zz.double().requires_grad_()
            (gd,) = torch.autograd.grad(torch.nn.functional.gelu(zd).sum(), zd)
            want = ref * gd
        else:
            r = torch.randn(M, N, generator=g).to(cuda)
            y = torch.randn(M, N, generator=g).to(cuda)
            y0 = y.double().clone()
            _gemm_batched(x, w, y, M, N, K, K, K, N, 0, 1, bias=bias, act=1, residual=r, res_pre=1, accumulate=1)
            want = y0 + torch.nn.functional.gelu(ref + bias.double() + r.double())
    torch.cuda.synchronize()
    tol = 1e-2 if y.dtype == torch.bfloat16 else 2e-5
    assert ((y.double() - want).abs().max() / want.abs().max()).item() <= tol


@pytest.mark.parametrize("M,N,K", [(3072, 768, 16200), (768, 3072, 4100), (1536, 1152, 2048), (200, 136, 5000), (128, 128, 1027)])
def test_weight_gradient_gemm_of_bf16_arrays(cuda, M, N, K):
    """gW = g^T x with both operands bf16 arrays (K = tokens, not a multiple of the K-step): split-K with float atomics into an
    fp32 gradient that already holds a value, and the bias gradient (column sums of g) as a by-product."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm
    gen = torch.Generator().manual_seed(M + K)
    g = torch.randn(K, M, generator=gen).to(cuda).to(torch.bfloat16)
    x = torch.randn(K, N, generator=gen).to(cuda).to(torch.bfloat16)
    gw = torch.randn(M, N, generator=gen).to(cuda)
    gb = torch.randn(M, generator=gen).to(cuda)
    want_w = gw.double() + g.double().T @ x.double()
    want_b = gb.double() + g.double().sum(0)
    with L.gemm_precision("bf16"):
        _gemm(g, x, gw, M, N, K, M, N, N, 1, 0, accumulate=1, rowsum=gb)
    torch.cuda.synchronize()
    assert ((gw.double() - want_w).abs().max() / want_w.abs().max()).item() <= 2e-5
    assert ((gb.double() - want_b).abs().max() / want_b.abs().max()).item() <= 2e-5


def test_clipping_norm_is_bit_reproducible(cuda):
    """dlwp_sumsq adds its block partials in index order: the same gradient gives the same bits every time (data-parallel ranks
    compute the clipping coefficient independently from the same all-reduced gradient)."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.randn(3_000_017, device=cuda)
    outs = []
    for _ in range(5):
        o = torch.zeros(1, device=cuda)
        L.check(lib.dlwp_sumsq(L.ptr(g), g.numel(), L.ptr(o), L.stream()))
        outs.append(o.item())
    assert len(set(outs)) == 1
    assert abs(outs[0] - float((g.double() ** 2).sum())) <= 1e-5 * outs[0]


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 192), (1000, 768, 64), (4100, 768, 1024), (2048, 2048, 4096)])
@pytest.mark.parametrize("epi", ["bias", "bias_gelu_preact_bf16", "residual_accumulate"])
def test_tile256_two_group_gemm_matches_float64(cuda, M, N, K, epi):
    """y = x W^T on the 256 x 256 x 64 kernel with the two wave groups half a phase apart (csrc/token_ops.hip, gemm_p8_kernel;
    forced on through lib.set_gemm_tile256): one K-tile up to sixteen, edge tiles in M and N, the epilogues of the token layers;
    repeated launches must agree bit for bit (a race between the LDS-DMA fills and the staggered readers would not)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(cuda).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda).to(torch.bfloat16)
    bias = torch.randn(N, generator=g).to(cuda)
    ref = x.double() @ w.double().T
    L.set_gemm_tile256(1)
    try:
        with L.gemm_precision("bf16"):
            outs = []
            for _ in range(3):
                if epi == "bias":          # (a product without any epilogue may be cut along K by the dispatcher: float atomics)
                    y = torch.empty(M, N, device=cuda)
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 0, None, None)
                    want = ref + bias.double()
                elif epi == "bias_gelu_preact_bf16":
                    y = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
                    z = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 1, z, None)
                    pre = ref + bias.double()
                    assert ((z.double() - pre).abs().max() / pre.abs().max()).item() <= 1e-2
                    want = torch.nn.functional.gelu(pre)
                else:
                    r = torch.randn(M, N, generator=torch.Generator().manual_seed(5)).to(cuda)
                    y = torch.ones(M, N, device=cuda)
                    _gemm_batched(x, w, y, M, N, K, K, K, N, 0, 1, bias=bias, residual=r, accumulate=1)
                    want = 1.0 + ref + bias.double() + r.double()
                torch.cuda.synchronize()
                outs.append(y.clone())
        tol = 1e-2 if y.dtype == torch.bfloat16 else 2e-5
        assert ((outs[0].double() - want).abs().max() / want.abs().max()).item() <= tol
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    finally:
        L.set_gemm_tile256(0)


@pytest.mark.parametrize("M,N,K", [(256, 256, 2048), (768, 512, 4100), (3072, 768, 16200), (520, 264, 3000)])
def test_tile256_weight_gradient_kernel(cuda, M, N, K):
    """gW = g^T x on the 256 x 256 two-group kernel (gemm_p8_tn_kernel, forced through lib.set_gemm_tile256): K slices to a slab,
    ordered reduction, a slice ending inside a K-tile, edge tiles, accumulation into an existing gradient, the bias gradient as a
    by-product; repeated launches agree bit for bit on the weight gradient (the slices are added in order)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm
    gen = torch.Generator().manual_seed(M + K)
    g = torch.randn(K, M, generator=gen).to(cuda).to(torch.bfloat16)
    x = torch.randn(K, N, generator=gen).to(cuda).to(torch.bfloat16)
    gw0 = torch.randn(M, N, generator=gen).to(cuda)
    gb0 = torch.randn(M, generator=gen).to(cuda)
    want_w = gw0.double() + g.double().T @ x.double()
    want_b = gb0.double() + g.double().sum(0)
    L.set_gemm_tile256(1)
    try:
        outs = []
        with L.gemm_precision("bf16"):
            for _ in range(3):
                gw, gb = gw0.clone(), gb0.clone()
                _gemm(g, x, gw, M, N, K, M, N, N, 1, 0, accumulate=1, rowsum=gb)
                torch.cuda.synchronize()
                outs.append(gw.clone())
        assert ((gw.double() - want_w).abs().max() / want_w.abs().max()).item() <= 2e-5
        assert ((gb.double() - want_b).abs().max() / want_b.abs().max()).item() <= 2e-5
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    finally:
        L.set_gemm_tile256(0)


@pytest.mark.parametrize("T,Nout,Kin", [(256, 128, 256), (1000, 192, 520), (4100, 1024, 768), (2048, 4096, 2048)])
def test_tile256_input_gradient_form(cuda, T, Nout, Kin):
    """gx = g W (B operand [k][n]: half-tile images [64 k][128], transposing reads) on the 256 x 256 two-group kernel, with the
    GELU'(z) multiply of the token-MLP backward in its epilogue; bit-reproducible over repeated launches."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm_batched
    gen = torch.Generator().manual_seed(T + Nout)
    g = torch.randn(T, Nout, generator=gen).to(cuda).to(torch.bfloat16)
    w = (torch.randn(Nout, Kin, generator=gen) / Nout ** 0.5).to(cuda).to(torch.bfloat16)
    z = torch.randn(T, Kin, generator=gen).to(cuda).to(torch.bfloat16)
    zd = z.double().requires_grad_()
    (gd,) = torch.autograd.grad(torch.nn.functional.gelu(zd).sum(), zd)
    want = (g.double() @ w.double()) * gd
    L.set_gemm_tile256(1)
    try:
        outs = []
        with L.gemm_precision("bf16"):
            for _ in range(3):
                gx = torch.empty(T, Kin, device=cuda, dtype=torch.bfloat16)
                _gemm_batched(g, w, gx, T, Kin, Nout, Nout, Kin, Kin, 0, 0, act=4, residual=z)
                torch.cuda.synchronize()
                outs.append(gx.clone())
        assert ((gx.double() - want).abs().max() / want.abs().max()).item() <= 1e-2
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    finally:
        L.set_gemm_tile256(0)


def test_weight_gradient_slab_survives_growth_behind_a_captured_graph(cuda):
    """The sliced weight-gradient GEMM's library-owned slab (csrc/token_ops.hip tn_slab_for): a hipGraph captured with a small
    product keeps working after a later, larger product has grown the slab (the old block is never freed), and a capture that would
    itself need a larger slab neither allocates nor synchronises (it takes the float-atomic epilogue) -- both replays reproduce
    the eager results."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    g = torch.Generator().manual_seed(4)

    def operands(M, N, K):
        return (torch.randn(K, M, generator=g).to(cuda).to(BF), torch.randn(K, N, generator=g).to(cuda).to(BF),
                torch.zeros(M, N, device=cuda))

    def run(gm, xm, out):
        K, M = gm.shape
        N = xm.shape[1]
        L.check(lib.dlwp_gemm_mixed(L.ptr(gm), L.ptr(xm), L.ptr(out), M, N, K, M, N, N, 1, 0, None, 0, None, None, 0, None, 3, L.stream()))
    with L.gemm_precision("bf16"):
        small, big = operands(1024, 768, 4096), operands(3072, 768, 8192)
        run(*small)                                   # eager warm-up sizes the slab for the small product
        torch.cuda.synchronize()
        want_small = small[2].clone()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        g_small, g_big = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.stream(s):
            with torch.cuda.graph(g_small):
                run(*small)
            free0 = torch.cuda.mem_get_info()[0]
            with torch.cuda.graph(g_big):             # needs more slab than exists: must not allocate while capturing
                run(*big)
            assert torch.cuda.mem_get_info()[0] == free0
        torch.cuda.current_stream().wait_stream(s)
        g_big.replay()
        torch.cuda.synchronize()
        big_captured = big[2].clone()
        run(*big)                                     # eager: grows the slab (a new block; the old one stays)
        torch.cuda.synchronize()
        ref_big = (big[0].float().t() @ big[1].float())
        assert ((big[2] - ref_big).abs().max() / ref_big.abs().max()).item() < 2e-3
        assert ((big_captured - ref_big).abs().max() / ref_big.abs().max()).item() < 2e-3
        g_small.replay()                              # recorded against the OLD slab
        torch.cuda.synchronize()
        assert torch.equal(small[2], want_small)


def _toy_run(cuda, mode):
    """One _ToyBlock under bf16 storage with gradient slots; mode: "full" (one ordinary backward), "partial" (torch.autograd.grad
    towards fc2's weight only: the qkv / proj nodes of the block never run), "twice" (two backward passes over a retained graph)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.train_engine import flatten_parameters, refresh_bf16_weights
    B, T, C, Hd = 2, 256, 64, 128
    g = torch.Generator().manual_seed(99)
    x0 = torch.randn(B, T, C, generator=g).to(cuda)
    gy = torch.randn(B, T, C, generator=g).to(cuda)
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        try:
            torch.manual_seed(1)
            m = _ToyBlock(C, Hd, 0.0).to(cuda).train()
            flatten_parameters(m)
            refresh_bf16_weights(m)
            L.SHADOW_ACTIVE = True
            try:
                y = m(x0.clone())
                if mode == "full":
                    y.backward(gy)
                elif mode == "partial":
                    torch.autograd.grad(y, [m.mlp.fc2.weight], grad_outputs=gy, allow_unused=True)
                else:
                    y.backward(gy, retain_graph=True)
                    y.backward(gy)
            finally:
                L.SHADOW_ACTIVE = False
        finally:
            L.set_storage("fp32")
    torch.cuda.synchronize()
    return {n: p.grad.clone() for n, p in m.named_parameters()}


def test_wgrad_batch_flushes_what_was_reported_when_a_backward_pass_skips_enrolled_nodes(cuda):
    """Advisor finding (round 5): the block's one weight-gradient launch waited for a counter to reach exactly zero; a pass in which
    an enrolled node never runs (torch.autograd.grad on a subset) silently dropped the gradients of the layers that DID report.
    Now the end-of-backward callback launches them."""
    full, part = _toy_run(cuda, "full"), _toy_run(cuda, "partial")
    for n in ("mlp.fc1.weight", "mlp.fc2.weight", "mlp.fc1.bias", "mlp.fc2.bias"):
        assert full[n].abs().sum().item() > 0
        assert (part[n] - full[n]).abs().max().item() <= 1e-5 * full[n].abs().max().item(), n
    assert part["qkv.weight"].abs().sum().item() == 0 and part["proj.weight"].abs().sum().item() == 0      # those nodes did not run


def test_wgrad_batch_is_rearmed_for_a_second_pass_over_a_retained_graph(cuda):
    full, twice = _toy_run(cuda, "full"), _toy_run(cuda, "twice")
    for n in full:          # slots accumulate: two passes leave twice the gradient
        assert (twice[n] - 2 * full[n]).abs().max().item() <= 2e-3 * full[n].abs().max().item() + 1e-12, n


@pytest.mark.parametrize("M,N,K", [(300, 520, 192), (4100, 768, 1024), (16200, 3072, 768)])
@pytest.mark.parametrize("form", ["fc1_gelu_stored_derivative", "gh_times_stored_derivative", "fc2_bias_residual_fp32", "plain_bf16"])
def test_tile256_wide_row_epilogue_is_the_narrow_one_bit_for_bit(cuda, M, N, K, form):
    """gemm_p8_kernel's register epilogue in 128-byte rows (round 6: v_permlane16_swap + masked DPP transposition of the C^T
    accumulators, epilogue_group8) against the round-5 form (8-byte pieces, DLWP_GEMM_P8_WIDE=0): the same element-wise arithmetic on
    the same accumulators, so every output -- activation, stored derivative, bf16 or fp32 -- must agree BIT FOR BIT, edge tiles
    included."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm, _gemm_batched
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(cuda).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(cuda).to(torch.bfloat16)
    wt = w.t().contiguous()                                                  # [K][N]: the "NN" operand form
    bias = torch.randn(N, generator=g).to(cuda)
    zd = torch.randn(M, N, generator=g).to(cuda).to(torch.bfloat16)
    res = torch.randn(M, N, generator=g).to(cuda)
    outs = {}
    L.set_gemm_tile256(1)
    try:
        with L.gemm_precision("bf16"):
            for wide in (1, 0):
                L.set_tuning("GEMM_P8_WIDE", wide)
                y = torch.full((M, N), 3.0, device=cuda, dtype=torch.float32 if form == "fc2_bias_residual_fp32" else torch.bfloat16)
                z = torch.full((M, N), 5.0, device=cuda, dtype=torch.bfloat16)
                if form == "fc1_gelu_stored_derivative":
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 7, z, None)
                elif form == "gh_times_stored_derivative":
                    _gemm_batched(x, wt, y, M, N, K, K, N, N, 0, 0, act=8, residual=zd)
                elif form == "fc2_bias_residual_fp32":
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1, bias, 0, None, res)
                else:
                    _gemm(x, w, y, M, N, K, K, K, N, 0, 1)
                torch.cuda.synchronize()
                outs[wide] = (y.clone(), z.clone())
    finally:
        L.set_tuning("GEMM_P8_WIDE", None)
        L.set_gemm_tile256(0)
    assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1])
    ref = x.double() @ w.double().T
    if form == "plain_bf16":
        assert ((outs[1][0].double() - ref).abs().max() / ref.abs().max()).item() <= 1e-2
