"""GPU parity of the window-attention path against golden vectors captured from the reference's own
classes (tests/golden/swin_golden.npz).  Tolerance: 1e-4 forward, 5e-4 gradients (fp32, max-norm)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "swin_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def load(module, prefix):
    sd = {k[len(prefix):]: t(k) for k in G.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all("relative_position_index" in m for m in missing), missing


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_window_attention_kernel_matches_reference_golden(cuda, tag):
    from dlwp_benchmark_amd.nsbench.swin_transformer import WindowAttention
    wa = WindowAttention(dim=12, window_size=(7, 7), num_heads=2)
    load(wa, "wa_p_")
    wa = wa.to(cuda)
    x = t(f"wa_{tag}_x").to(cuda).requires_grad_(True)
    labels = torch.from_numpy(G["wa_labels"]).to(cuda) if tag == "mask" else None
    y = wa(x, labels, 4)
    assert rel(y, t(f"wa_{tag}_y")) <= 1e-4
    y.backward(t(f"wa_{tag}_gy").to(cuda))
    assert rel(x.grad, t(f"wa_{tag}_gx")) <= 5e-4
    for n, p in wa.named_parameters():
        assert rel(p.grad, t(f"wa_{tag}_g_{n}")) <= 5e-4, n


@pytest.mark.parametrize("tag,H,W,pm", [("28x28", 28, 28, "constant"), ("20x30", 20, 30, "circular")])
def test_basic_layer_window7_matches_reference_golden(cuda, tag, H, W, pm):
    from dlwp_benchmark_amd.nsbench.swin_transformer import BasicLayer
    bl = BasicLayer(dim=8, depth=2, num_heads=2, window_size=7, padding_mode=pm)
    load(bl, f"bl_{tag}_p_")
    bl = bl.to(cuda)
    x = t(f"bl_{tag}_x").to(cuda).requires_grad_(True)
    y = bl(x, H, W)[0]
    assert rel(y, t(f"bl_{tag}_y")) <= 1e-4
    y.backward(t(f"bl_{tag}_gy").to(cuda))
    assert rel(x.grad, t(f"bl_{tag}_gx")) <= 5e-4
    for n, p in bl.named_parameters():
        assert rel(p.grad, t(f"bl_{tag}_g_{n}")) <= 1e-3, n


def test_swin_rollout_matches_reference_golden(cuda):
    from dlwp_benchmark_amd.nsbench.swin_transformer import SwinTransformer
    net = SwinTransformer(context_size=2, pretrain_img_size=32, patch_size=2, in_chans=1, out_chans=1, embed_dim=8,
                          depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, type="SwinTransformer", name="t")
    load(net, "net_p_")
    net = net.to(cuda)
    y = net(t("net_x").to(cuda), teacher_forcing_steps=2)
    assert rel(y, t("net_y")) <= 1e-4
    loss = torch.nn.functional.mse_loss(y, t("net_target").to(cuda))
    assert abs(loss.item() - float(G["net_loss"])) <= 1e-4 * abs(float(G["net_loss"]))
    loss.backward()
    for n, p in net.named_parameters():
        if "net_g_" + n in G.files:
            assert rel(p.grad, t("net_g_" + n)) <= 2e-3, n


def test_large_window_many_tiles_matches_torch(cuda):
    """N = 1024 tokens (window = whole 32x32 map as at nsbench stage 0), head_dim 10: 16 key tiles."""
    from dlwp_benchmark_amd.nsbench.swin_transformer import WindowAttention
    from oracle import swin_ref
    g = torch.Generator().manual_seed(8)
    wa = WindowAttention(dim=40, window_size=(32, 32), num_heads=4)
    with torch.no_grad():
        wa.relative_position_bias_table.copy_(torch.randn(wa.relative_position_bias_table.shape, generator=g) * 0.5)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in wa.named_parameters()}
    x = torch.randn(2, 1024, 40, generator=g)
    labels = torch.randint(0, 4, (1, 1024), generator=g).to(torch.int32)
    gy = torch.randn(2, 1024, 40, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = swin_ref.window_attention(xr, p, "", 32, 32, 4, labels)
    yr.backward(gy)
    wa = wa.to(cuda)
    xd = x.to(cuda).requires_grad_(True)
    y = wa(xd, labels.to(cuda), 1)
    y.backward(gy.to(cuda))
    assert rel(y, yr) <= 1e-4
    assert rel(xd.grad, xr.grad) <= 5e-4
    for n, q in wa.named_parameters():
        assert rel(q.grad, p[n].grad) <= 1e-3, n


@pytest.mark.parametrize("Wh,Ww,dim,heads,B_,nW", [(7, 7, 192, 4, 6, 3), (16, 32, 192, 4, 2, 1), (8, 8, 256, 4, 4, 2), (5, 9, 80, 2, 3, 1),
                                                    # wide heads of the deep nsbench U-Net stages (GEMM form): the shipped
                                                    # swintransformer.yaml reaches 384 / 4 = 96 on 8 x 8 and 768 / 4 = 192 on
                                                    # 4 x 4 tokens; the paper's 4-stage runs 112 (embed 56)
                                                    (8, 8, 384, 4, 4, 1), (4, 4, 768, 4, 4, 1), (4, 4, 448, 4, 6, 2),
                                                    (10, 13, 144, 2, 2, 2)])
def test_head_dims_up_to_64_match_oracle(cuda, Wh, Ww, dim, heads, B_, nW):
    """head_dim 40 / 48 / 64 (three and four 16-wide MFMA blocks per row: dlwpbench Swin stage 1 has 192 / 4 = 48) on the
    fused kernels, and 72 ... 192 on the GEMM form."""
    from dlwp_benchmark_amd.nsbench.swin_transformer import WindowAttention
    from oracle import swin_ref
    g = torch.Generator().manual_seed(18)
    N = Wh * Ww
    wa = WindowAttention(dim=dim, window_size=(Wh, Ww), num_heads=heads)
    with torch.no_grad():
        wa.relative_position_bias_table.copy_(torch.randn(wa.relative_position_bias_table.shape, generator=g) * 0.5)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in wa.named_parameters()}
    x = torch.randn(B_, N, dim, generator=g)
    labels = torch.randint(0, 3, (nW, N), generator=g).to(torch.int32)
    gy = torch.randn(B_, N, dim, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = swin_ref.window_attention(xr, p, "", Wh, Ww, heads, labels)
    yr.backward(gy)
    wa = wa.to(cuda)
    xd = x.to(cuda).requires_grad_(True)
    y = wa(xd, labels.to(cuda), nW)
    y.backward(gy.to(cuda))
    assert rel(y, yr) <= 1e-4
    assert rel(xd.grad, xr.grad) <= 5e-4
    for n, q in wa.named_parameters():
        assert rel(q.grad, p[n].grad) <= 1e-3, n


# ---- dlwpbench twin --------------------------------------------------------------------------------------
GD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dlwp_swin_golden.npz"))
DLWP_CFG = {"one": dict(constant_channels=2, prescribed_channels=1, prognostic_channels=3, context_size=1, img_height=16,
                        img_width=32, patch_size=2, embed_dim=8, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0),
            "multi": dict(constant_channels=2, prescribed_channels=1, prognostic_channels=2, context_size=2, img_height=16,
                          img_width=32, patch_size=1, embed_dim=8, depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0)}


@pytest.mark.parametrize("tag", ["one", "multi"])
def test_dlwp_swin_matches_reference_golden(cuda, tag):
    from dlwp_benchmark_amd import dlwpbench
    td = lambda n: torch.from_numpy(GD[f"{tag}_{n}"])   # noqa: E731
    net = dlwpbench.SwinTransformer(**DLWP_CFG[tag])
    sd = {k[len(tag) + 3:]: torch.from_numpy(GD[k]) for k in GD.files if k.startswith(f"{tag}_p_")}
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("relative_position_index" in m for m in missing), (missing, unexpected)
    net = net.to(cuda).train()
    y = net(constants=td("constants").to(cuda), prescribed=td("prescribed").to(cuda), prognostic=td("prognostic").to(cuda))
    assert rel(y, td("y")) <= 1e-4
    loss = torch.nn.functional.mse_loss(y, td("target").to(cuda))
    assert abs(loss.item() - float(GD[f"{tag}_loss"])) <= 1e-4 * abs(float(GD[f"{tag}_loss"]))
    loss.backward()
    for n, p in net.named_parameters():
        if f"{tag}_g_{n}" in GD.files:
            assert rel(p.grad, td(f"g_{n}")) <= 2e-3, n


@pytest.mark.parametrize("Wh,Ww,pl,d,heads,B_,nW,ntypes,masked", [(7, 7, 1, 24, 4, 640, 20, 1, True), (7, 7, 2, 32, 6, 360, 30, 5, True),
                                                                   (7, 7, 1, 10, 4, 600, 12, 1, False), (8, 8, 1, 16, 8, 300, 4, 2, True)])
def test_many_short_windows_wave_kernels_match_tiled_kernels(cuda, monkeypatch, Wh, Ww, pl, d, heads, B_, nW, ntypes, masked):
    """Above 2048 (window, head) pairs windows of at most 128 tokens take the wave-per-window kernels (csrc/winattn_small.hip);
    DLWP_WINATTN_TILED=1 forces the tiled kernels, which the goldens above pin: both must agree, forward and backward."""
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_core
    assert B_ * heads >= 2048
    N = pl * Wh * Ww
    TB = 3 * N
    g = torch.Generator().manual_seed(N + d + heads)
    qkv0 = torch.randn(B_, N, 3 * heads * d, generator=g).to(cuda)
    table0 = (0.5 * torch.randn(TB, ntypes, heads, generator=g)).to(cuda)
    if ntypes == 1:
        table0 = table0[:, 0].contiguous()
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(cuda) if masked else None
    gy = torch.randn(B_, N, heads * d, generator=g).to(cuda)
    res = []
    for tiled in (True, False):
        if tiled:
            monkeypatch.setenv("DLWP_WINATTN_TILED", "1")
        else:
            monkeypatch.delenv("DLWP_WINATTN_TILED", raising=False)
        qkv, table = qkv0.clone().requires_grad_(), table0.clone().requires_grad_()
        y = window_attention_core(qkv, table, ia, ib, labels, nW, heads, d ** -0.5)
        y.backward(gy)
        res.append((y.detach(), qkv.grad, table.grad))
    assert rel(res[1][0], res[0][0]) <= 2e-5
    assert rel(res[1][1], res[0][1]) <= 1e-4
    assert rel(res[1][2], res[0][2]) <= 1e-4


@pytest.mark.parametrize("N_side,d,heads,B_,nW", [(7, 24, 4, 16, 4), (7, 24, 4, 640, 20), (12, 32, 2, 3, 1), (16, 16, 4, 2, 2)])
def test_window_attention_bf16_matrix_arithmetic_stays_close(cuda, N_side, d, heads, B_, nW):
    """dlwp_set_gemm_precision(1): Q K^T, P V and the backward products run on v_mfma_f32_16x16x16_bf16 (bf16 operands, fp32
    accumulation; softmax, bias and statistics fp32) in the tiled kernels and in the wave-per-window kernels.  The result
    must stay within bf16 rounding of the fp32 kernels (the reference under autocast does these matmuls in bf16)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_core
    N = N_side * N_side
    TB = 3 * N
    g = torch.Generator().manual_seed(N + d)
    qkv0 = torch.randn(B_, N, 3 * heads * d, generator=g).to(cuda)
    table0 = (0.5 * torch.randn(TB, heads, generator=g)).to(cuda)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(cuda)
    gy = torch.randn(B_, N, heads * d, generator=g).to(cuda)
    res = []
    for mode in ("fp32", "bf16"):
        with L.gemm_precision(mode):
            qkv, table = qkv0.clone().requires_grad_(), table0.clone().requires_grad_()
            y = window_attention_core(qkv, table, ia, ib, labels, nW, heads, d ** -0.5)
            y.backward(gy)
        res.append((y.detach(), qkv.grad, table.grad))
    assert rel(res[1][0], res[0][0]) <= 2e-2
    assert rel(res[1][1], res[0][1]) <= 3e-2
    assert rel(res[1][2], res[0][2]) <= 3e-2
    assert not torch.equal(res[1][0], res[0][0])          # the bf16 path really ran


@pytest.mark.parametrize("Wh,Ww,pl,d,heads,B_,nW,ntypes,masked,qrange", [(7, 7, 1, 24, 4, 640, 20, 1, True, None),
                                                                          (7, 7, 2, 32, 6, 360, 30, 5, True, None),
                                                                          (7, 7, 2, 32, 6, 360, 30, 5, True, (49, 98)),
                                                                          (7, 7, 2, 32, 6, 360, 30, 5, False, (0, 49)),
                                                                          (8, 8, 1, 16, 8, 300, 4, 2, True, None)])
def test_lds_staged_bf16_window_kernels_match_the_register_fragment_kernels(cuda, monkeypatch, Wh, Ww, pl, d, heads, B_, nW, ntypes,
                                                                            masked, qrange):
    """bf16 matrix mode, many short windows: a workgroup stages one (window, head) in LDS as bf16 (csrc/winattn_small.hip,
    winattn_lds_*); same products, order and roundings as the register-fragment wave kernels (DLWP_WINATTN_NOLDS=1) -- with a
    query range the rows outside it are undefined in `out` and must be zero in the query gradient."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_core
    N = pl * Wh * Ww
    TB = 3 * N
    g = torch.Generator().manual_seed(N + d + heads)
    qkv0 = torch.randn(B_, N, 3 * heads * d, generator=g).to(cuda)
    table0 = (0.5 * torch.randn(TB, ntypes, heads, generator=g)).to(cuda)
    if ntypes == 1:
        table0 = table0[:, 0].contiguous()
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(cuda) if masked else None
    gy = torch.randn(B_, N, heads * d, generator=g).to(cuda)
    lo, hi = (0, N) if qrange is None else qrange
    clo, chi = lo // 16 * 16, min(N, (hi + 15) // 16 * 16)          # whole chunks of 16 query rows are computed
    gy[:, :clo] = 0
    gy[:, chi:] = 0                                                # rows outside the range carry no upstream gradient
    res = []
    with L.gemm_precision("bf16"):
        for nolds in (True, False):
            if nolds:
                monkeypatch.setenv("DLWP_WINATTN_NOLDS", "1")
            else:
                monkeypatch.delenv("DLWP_WINATTN_NOLDS", raising=False)
            qkv, table = qkv0.clone().requires_grad_(), table0.clone().requires_grad_()
            y = window_attention_core(qkv, table, ia, ib, labels, nW, heads, d ** -0.5, qrange)
            y.backward(gy)
            res.append((y.detach()[:, clo:chi], qkv.grad, table.grad))
    # D = rowsum(dO o) is summed in another order (8 threads per row at staging): an fp32 ulp there moves a bf16 rounding of dS here
    # and there -- 3e-4 observed; the bf16-vs-fp32 distance of either family is 100x that (test above)
    assert rel(res[1][0], res[0][0]) <= 1e-5
    assert rel(res[1][1], res[0][1]) <= 1e-3
    assert rel(res[1][2], res[0][2]) <= 1e-3
    if qrange is not None:
        q_part = res[1][1].reshape(B_, N, 3, heads * d)[:, :, 0]
        assert float(q_part[:, :clo].abs().max() if clo else 0.0) == 0.0 and float(q_part[:, chi:].abs().max() if chi < N else 0.0) == 0.0


@pytest.mark.parametrize("d,heads,B_,nW,masked", [(48, 4, 380, 190, True), (40, 2, 600, 12, False), (36, 4, 512, 1, True)])
def test_head_dims_33_to_48_on_the_wave_and_lds_families_in_bf16_mode(cuda, monkeypatch, d, heads, B_, nW, masked):
    """Round 6: windows of at most 64 tokens with head dims 33 .. 48 (Swin C4 stage 1: 192 channels on 4 heads, 49 tokens, 1520 (window, head)
    pairs at batch 2) take the wave-per-window forward (winattn_small_fwd_kernel<4, 3>) and the LDS-staged two-pass backward
    (winattn_lds_bwd_kernel<3>, 104-byte rows) in the bf16 matrix mode instead of the tiled kernels.  Against the tiled kernels in the same
    arithmetic (DLWP_WINATTN_D48=0; both round operands and probabilities to bf16, sums in fp32: 1e-2) and against the fp32 tiled kernels
    (bf16 distance: 2e-2 / 3e-2, as test_window_attention_bf16_matrix_arithmetic_stays_close)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_core
    assert B_ * heads >= 1024
    N, TB = 49, 3 * 49
    g = torch.Generator().manual_seed(d + heads)
    qkv0 = torch.randn(B_, N, 3 * heads * d, generator=g).to(cuda)
    table0 = (0.5 * torch.randn(TB, heads, generator=g)).to(cuda)
    ia = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    ib = torch.randint(0, TB // 2, (N,), generator=g, dtype=torch.int32).to(cuda)
    labels = torch.randint(0, 3, (nW, N), generator=g, dtype=torch.int32).to(cuda) if masked else None
    gy = torch.randn(B_, N, heads * d, generator=g).to(cuda)

    def run(mode, d48):
        monkeypatch.setenv("DLWP_WINATTN_D48", "1" if d48 else "0")
        with L.gemm_precision(mode):
            qkv, table = qkv0.clone().requires_grad_(), table0.clone().requires_grad_()
            with L.kernel_accounting() as acc:
                y = window_attention_core(qkv, table, ia, ib, labels, nW, heads, d ** -0.5)
                y.backward(gy)
                torch.cuda.synchronize()
        return (y.detach(), qkv.grad, table.grad), {r["name"] for r in acc.rows}
    ref32, _ = run("fp32", False)
    tiled, names_t = run("bf16", False)
    new, names_n = run("bf16", True)
    monkeypatch.delenv("DLWP_WINATTN_D48", raising=False)
    assert any(n.startswith("winattn_lds_bwd_kernel<3>") for n in names_n) and any(n.startswith("winattn_small_fwd_kernel") for n in names_n), names_n
    assert any(n.startswith("winattn_bwd_q_kernel") for n in names_t), names_t
    for i, tol in enumerate((1e-2, 1e-2, 1e-2)):
        assert rel(new[i], tiled[i]) <= tol, (i, rel(new[i], tiled[i]))
    for i, tol in enumerate((2e-2, 3e-2, 3e-2)):
        assert rel(new[i], ref32[i]) <= tol, (i, rel(new[i], ref32[i]))


@pytest.mark.parametrize("H,W,pm,shift", [(20, 30, ("constant", "circular"), 3), (20, 30, ("constant", "circular"), 0),
                                          (28, 28, "constant", 3), (21, 35, ("constant", "circular"), 3)])
@pytest.mark.parametrize("bias", [True, False])
def test_real_token_flow_equals_window_token_flow(cuda, H, W, pm, shift, bias):
    """SwinTransformerBlock: qkv on the real tokens + partition of the projected tensor (circular copies; constant pads filled with
    the qkv bias) + proj after reverse must equal the reference order (pad / roll / partition, then qkv and proj on every window
    token: src/nsbench/models/swintransformer/swin_transformer.py:229-250) -- output and every gradient, fp32.  The reference-order
    flow itself is pinned by the goldens above."""
    from dlwp_benchmark_amd.nsbench.swin_transformer import SwinTransformerBlock
    torch.manual_seed(9)
    dim, heads, B = 32, 2, 2
    blk = SwinTransformerBlock(dim, heads, window_size=7, shift_size=shift, qkv_bias=bias, padding_mode=pm).to(cuda)
    blk.H, blk.W = H, W
    with torch.no_grad():
        blk.attn.relative_position_bias_table.normal_(0, 0.5)
        if bias:
            blk.attn.qkv.bias.normal_(0, 0.5)
    labels = None
    if shift:
        import math
        from dlwp_benchmark_amd.nsbench.swin_transformer import BasicLayer
        layer = BasicLayer(dim, depth=2, num_heads=heads, window_size=7, padding_mode=pm)
        labels = layer.shift_labels(math.ceil(H / 7) * 7, math.ceil(W / 7) * 7, cuda)
    x = torch.randn(B, H * W, dim, device=cuda, requires_grad=True)
    g = torch.randn(B, H * W, dim, device=cuda)
    outs = {}
    for flow in (True, False):
        blk.real_token_flow = flow
        y = blk(x, labels)
        outs[flow] = (y.detach().clone(), torch.autograd.grad(y, [x] + list(blk.parameters()), g))
    assert rel(outs[True][0], outs[False][0]) <= 2e-5
    for n, a, b in zip(["x"] + [n for n, _ in blk.named_parameters()], outs[True][1], outs[False][1]):
        assert rel(a, b) <= 2e-4, n


@pytest.mark.parametrize("storage", ["fp32", "bf16"])
@pytest.mark.parametrize("pm", ["constant", ("constant", "circular")])
def test_swin_blocks_with_fused_stochastic_depth_and_bf16_window_moves_equal_the_separate_passes(cuda, storage, pm):
    """BasicLayer (an unshifted and a shifted SwinTransformerBlock) in training mode with drop_path > 0: the per-sample scale inside the scatter kernel / fc2's epilogue, bf16
    rows through the gather / scatter under bf16 storage and the four weight gradients in one launch, against the same block with
    DLWP_DROPPATH_FUSED / DLWP_WGRAD_BATCH off (separate scale passes, per-layer products), same masks: forward and every gradient.
    fp32 storage: 1e-5 of the max norm; bf16 storage: 2e-2 (bf16 roundings at different places of the backward chain)."""
    from dlwp_benchmark_amd import lib as L, token_ops as TO
    from dlwp_benchmark_amd.nsbench.swin_transformer import BasicLayer
    from dlwp_benchmark_amd.train_engine import flatten_parameters, refresh_bf16_weights
    B, H, W, C = 3, 12, 20, 32
    g = torch.Generator().manual_seed(5)
    x0 = torch.randn(B, H * W, C, generator=g).to(cuda)
    gy = torch.randn(B, H * W, C, generator=g).to(cuda)
    masks = [torch.tensor(m, device=cuda) for m in ([0.0, 1.25, 1.25], [1.25, 0.0, 1.25], [1.25, 1.25, 0.0], [1.25, 1.25, 1.25])]
    res = {}

    def run(fused):
        torch.manual_seed(2)
        blk = BasicLayer(C, depth=2, num_heads=4, window_size=7, mlp_ratio=2.0, drop_path=0.2, padding_mode=pm).to(cuda).train()
        flat, grad = flatten_parameters(blk)
        refresh_bf16_weights(blk)
        it = iter(masks)
        for b_ in blk.blocks:          # (unshifted block, shifted block) x (attention branch, MLP branch)
            b_.drop_path.mask = lambda batch, device: next(it)
        TO.DROPPATH_FUSED = TO.WGRAD_BATCH = fused
        L.SHADOW_ACTIVE = True
        try:
            x = x0.clone().requires_grad_(True)
            y = blk(x, H, W)[0]
            y.backward(gy)
        finally:
            L.SHADOW_ACTIVE = False
            TO.DROPPATH_FUSED = TO.WGRAD_BATCH = True
        return [y.detach(), x.grad] + [p.grad.clone() for p in blk.parameters()]
    ctx = L.gemm_precision("bf16") if storage == "bf16" else __import__("contextlib").nullcontext()
    with ctx:
        if storage == "bf16":
            L.set_storage("bf16")
        try:
            a, b = run(True), run(False)
        finally:
            L.set_storage("fp32")
    tol = 1e-5 if storage == "fp32" else 2e-2
    for u, v in zip(a, b):
        assert (u - v).abs().max().item() <= tol * v.abs().max().item() + 1e-12


# ---- ape=True (absolute position embedding; fixtures: tests/golden/make_swin_ape_golden.py) ----------------------------
GA = np.load(os.path.join(os.path.dirname(__file__), "golden", "swin_ape_golden.npz"))


def _load_ape(net, tag):
    sd = {k[len(tag) + 3:]: torch.from_numpy(GA[k]) for k in GA.files if k.startswith(f"{tag}_p_")}
    assert "absolute_pos_embed" in sd
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all("relative_position_index" in m for m in missing), (missing, unexpected)


@pytest.mark.parametrize("tag", ["ns", "ns_resized"])
def test_ns_swin_ape_matches_reference_golden(cuda, tag):
    """`ns`: frame = pretraining size; `ns_resized`: 24 x 24 frames on the 16 x 16 embedding (the bicubic resize is real)."""
    from dlwp_benchmark_amd.nsbench.swin_transformer import SwinTransformer
    ta = lambda n: torch.from_numpy(GA[f"{tag}_{n}"])   # noqa: E731
    net = SwinTransformer(context_size=2, pretrain_img_size=32, patch_size=2, in_chans=1, out_chans=1, embed_dim=8,
                          depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0, ape=True)
    _load_ape(net, tag)
    net = net.to(cuda)
    y = net(ta("x").to(cuda), teacher_forcing_steps=2)
    assert rel(y, ta("y")) <= 1e-4
    loss = torch.nn.functional.mse_loss(y, ta("target").to(cuda))
    loss.backward()
    for n, p in net.named_parameters():
        if f"{tag}_g_{n}" in GA.files:
            assert rel(p.grad, ta(f"g_{n}")) <= 2e-3, n


def test_dlwp_swin_ape_matches_reference_golden(cuda):
    from dlwp_benchmark_amd import dlwpbench
    ta = lambda n: torch.from_numpy(GA[f"dlwp_{n}"])   # noqa: E731
    net = dlwpbench.SwinTransformer(**DLWP_CFG["multi"], ape=True)
    _load_ape(net, "dlwp")
    net = net.to(cuda).train()
    y = net(constants=ta("constants").to(cuda), prescribed=ta("prescribed").to(cuda), prognostic=ta("prognostic").to(cuda))
    assert rel(y, ta("y")) <= 1e-4
    loss = torch.nn.functional.mse_loss(y, ta("target").to(cuda))
    loss.backward()
    for n, p in net.named_parameters():
        if f"dlwp_g_{n}" in GA.files:
            assert rel(p.grad, ta(f"g_{n}")) <= 2e-3, n
    # the bf16 activation mode keeps the embedding add in the activation dtype
    assert net.absolute_pos_embed.grad is not None and net.absolute_pos_embed.grad.shape == net.absolute_pos_embed.shape
