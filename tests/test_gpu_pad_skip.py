"""Pangu's padded window tokens (src/dlwpbench/models/panguweather/panguweather.py:283-317: ZeroPad3d after norm1, qkv / attention
/ proj on every window token, crop3d): the real-token flow -- qkv on the real tokens, the qkv tensor padded with the qkv bias,
padded tokens as keys / values only, proj after the crop -- must equal the reference-order flow exactly, forward and backward."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("C", [8, 1152])          # 1152 = qkv width of Pangu's deep stages: more channel quads than threads
def test_partition_with_fill_and_its_adjoint(cuda, C):
    from dlwp_benchmark_amd.window_ops import WindowSpec, partition
    spec = WindowSpec((1, 9, 12), (2, 4, 5), front=(0, 1, 1), back=(1, 2, 2), order=(2, 0, 1))
    torch.manual_seed(0)
    B = 2
    x = torch.randn(B, 9 * 12, C, device=cuda, requires_grad=True)
    fill = torch.randn(C, device=cuda, requires_grad=True)
    for shift in ((0, 0, 0), (1, 2, 2)):
        plain = partition(x, spec, shift)
        pads = partition(torch.ones_like(x), spec, shift)[..., :1] == 0          # [B*nW, N, 1]
        got = partition(x, spec, shift, fill=fill)
        want = torch.where(pads, fill.expand_as(plain), plain)
        assert torch.equal(got, want)
        g = torch.randn_like(got)
        gx, gf = torch.autograd.grad(got, (x, fill), g)
        gx_ref, gf_ref = torch.autograd.grad(want, (x, fill), g)
        assert rel(gx, gx_ref) <= 1e-6 and rel(gf, gf_ref) <= 1e-5
        assert int(pads.sum()) > 0


@pytest.mark.parametrize("modes", [("constant",) * 3, ("constant", "constant", "circular")])
def test_partition_and_reverse_on_bf16_arrays_with_the_branch_scale(cuda, modes):
    """dlwp_window_gather_ex / dlwp_window_scatter_ex: a bf16 array moves through partition bit for bit (and its adjoint sums the
    copies in fp32, rounded once); reverse(bf16 windows, residual, row_scale) = residual + scale[b] * reverse(fp32 windows) within
    one bf16 rounding of the windows, with the gradient of the windows = scale[b] * gather(g) in the windows' dtype."""
    from dlwp_benchmark_amd.window_ops import WindowSpec, partition, reverse
    BF = torch.bfloat16
    spec = WindowSpec((1, 9, 12), (1, 4, 5), front=(0, 1, 1), back=(0, 2, 2), modes=modes)
    torch.manual_seed(1)
    B, C = 3, 24
    for shift in ((0, 0, 0), (0, 2, 2)):
        x = torch.randn(B, 9 * 12, C, device=cuda)
        x16 = x.to(BF).requires_grad_(True)
        x32 = x.to(BF).float().requires_grad_(True)
        w16, w32 = partition(x16, spec, shift), partition(x32, spec, shift)
        assert w16.dtype == BF and torch.equal(w16.float(), w32)
        g = torch.randn_like(w32)
        (g16,), (g32,) = torch.autograd.grad(w16, x16, g.to(BF)), torch.autograd.grad(w32, x32, g.to(BF).float())
        assert g16.dtype == BF and torch.equal(g16, g32.to(BF))
        # reverse with scale + residual
        wins = torch.randn(B * spec.nW, spec.N, C, device=cuda)
        res = torch.randn(B, 9 * 12, C, device=cuda, requires_grad=True)
        sc = torch.tensor([0.0, 1.25, 1.0], device=cuda)
        a16 = wins.to(BF).requires_grad_(True)
        a32 = wins.to(BF).float().requires_grad_(True)
        y16 = reverse(a16, spec, B, shift, residual=res, row_scale=sc)
        y32 = res + sc[:, None, None] * reverse(a32, spec, B, shift)
        assert y16.dtype == torch.float32 and rel(y16, y32) <= 1e-6
        gy = torch.randn_like(y32)
        ga16, gr16 = torch.autograd.grad(y16, (a16, res), gy)
        ga32, gr32 = torch.autograd.grad(y32, (a32, res), gy)
        assert ga16.dtype == BF and torch.equal(ga16, ga32.to(BF)) and torch.equal(gr16, gr32)
        assert torch.equal(y16[0], res[0].detach())          # a dropped sample is its residual


@pytest.mark.parametrize("B,res,heads,dim", [(8, (1, 30, 60), 6, 96),      # 8 * 45 windows * 6 heads: the wave-per-window kernels
                                              (1, (1, 30, 60), 6, 96),      # few windows: the tiled kernels (range ignored)
                                              (2, (2, 20, 30), 4, 64)])     # no pressure-level pad: lat / lon pads only
@pytest.mark.parametrize("shifted", [False, True])
def test_real_token_flow_equals_reference_order(cuda, B, res, heads, dim, shifted):
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    torch.manual_seed(5)
    blk = EarthSpecificBlock(dim, res, heads, (2, 7, 7), None if shifted else (0, 0, 0)).to(cuda)
    with torch.no_grad():
        blk.attn.earth_position_bias_table.normal_(0, 0.5)
        blk.attn.qkv.bias.normal_(0, 0.5)
    L_ = res[0] * res[1] * res[2]
    x = torch.randn(B, L_, dim, device=cuda, requires_grad=True)
    g = torch.randn(B, L_, dim, device=cuda)
    outs = {}
    for flow in (False, True):
        blk.real_token_flow = flow
        blk.zero_grad(set_to_none=True)
        y = blk(x)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), g)
        outs[flow] = (y.detach().clone(), [t.clone() for t in grads])
    assert rel(outs[True][0], outs[False][0]) <= 2e-5
    names = ["x"] + [n for n, _ in blk.named_parameters()]
    for n, a, b in zip(names, outs[True][1], outs[False][1]):
        assert rel(a, b) <= 2e-4, n
    if B > 2:
        return
    # ... and BOTH equal the CPU restatement of the reference block (oracle/pangu_ref.earth_block, pinned by
    # tests/golden/pangu_golden.npz in tests/test_oracle_pangu.py), evaluated in float64 with autograd: output and every gradient
    from oracle import pangu_ref
    p64 = {n: t.detach().double().cpu().requires_grad_(True) for n, t in blk.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = pangu_ref.earth_block(x64, p64, "", res, heads, (2, 7, 7), pangu_ref.DEFAULT_SHIFT if shifted else (0, 0, 0))
    g64 = torch.autograd.grad(y64, [x64] + [p64[n] for n in names[1:]], g.double().cpu())
    for flow in (False, True):
        assert rel(outs[flow][0], y64) <= 2e-5, flow
        for n, a, b in zip(names, outs[flow][1], g64):
            assert rel(a, b) <= 2e-4, (flow, n)


@pytest.mark.parametrize("B,res,heads,dim", [(2, (1, 30, 60), 6, 96), (1, (2, 20, 30), 4, 64), (8, (1, 30, 60), 6, 192)])
@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("fused_fwd", [True, False])
def test_token_layout_backward_equals_the_four_launch_chain(cuda, monkeypatch, B, res, heads, dim, shifted, fused_fwd):
    """bf16 matrix mode: partition + attention + reverse as one node with the one-launch backward (dlwp_window_attn_bwd_tokens:
    gradients through the position maps, padded positions summed into the qkv bias) against the separate nodes (gather of gout,
    dlwp_window_attn_bwd_qrange, scatter, pad column sum) -- the same kernel arithmetic: 2e-5 on the input gradient, 1e-4 on the
    parameter gradients (column sums over all tokens: float atomics reorder them) -- and against the float64 reference block at
    the bf16-operand tolerance."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    from dlwp_benchmark_amd.nsbench import swin_transformer as st
    from oracle import pangu_ref
    torch.manual_seed(7)
    blk = EarthSpecificBlock(dim, res, heads, (2, 7, 7), None if shifted else (0, 0, 0)).to(cuda)
    with torch.no_grad():
        blk.attn.earth_position_bias_table.normal_(0, 0.5)
        blk.attn.qkv.bias.normal_(0, 0.5)
    blk.real_token_flow = True
    L_ = res[0] * res[1] * res[2]
    x = torch.randn(B, L_, dim, device=cuda, requires_grad=True)
    g = torch.randn(B, L_, dim, device=cuda)
    names = ["x"] + [n for n, _ in blk.named_parameters()]
    outs = {}
    monkeypatch.setattr(st, "FUSED_FWD", fused_fwd)      # True: dlwp_window_attn_fwd_tokens where it applies (the 2160-pair case)
    with L.gemm_precision("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(st, "FUSED_BWD", fused)
            assert st._WindowAttnTokensFn.applies(x, blk._wspec, dim // heads, blk.attn.earth_position_bias_table) == fused
            y = blk(x)
            grads = torch.autograd.grad(y, [x] + list(blk.parameters()), g)
            outs[fused] = (y.detach().clone(), [t.clone() for t in grads])
    assert rel(outs[True][0], outs[False][0]) <= 1e-6          # (the token-layout forward is the same kernel on other addresses)
    # with fewer than 2048 (window, head) pairs the separate attention node runs the TILED backward kernels (other summation
    # order on bf16 operands): compared at 1e-3 there, tightly where both sides run the wave-per-window family
    same_kernel = B * blk._wspec.nW * heads >= 2048
    for n, a, b in zip(names, outs[True][1], outs[False][1]):
        assert rel(a, b) <= ((2e-5 if n == "x" else 1e-4) if same_kernel else 1e-3), n
    p64 = {n: t.detach().double().cpu().requires_grad_(True) for n, t in blk.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = pangu_ref.earth_block(x64, p64, "", res, heads, (2, 7, 7), pangu_ref.DEFAULT_SHIFT if shifted else (0, 0, 0))
    g64 = torch.autograd.grad(y64, [x64] + [p64[n] for n in names[1:]], g.double().cpu())
    assert rel(outs[True][0], y64) <= 2e-2
    for n, a, b in zip(names, outs[True][1], g64):
        assert rel(a, b) <= 3e-2, n


@pytest.mark.parametrize("shifted", [False, True])
def test_bf16_token_tensors_under_bf16_storage(cuda, shifted):
    """bf16 storage live (the engine's state): the qkv projection writes bf16, the attention kernels read / write bf16 token rows
    (dlwp_window_attn_fwd_tokens / _bwd_tokens with io_bf16) and proj reads them; output and every gradient against the float64
    reference block at the bf16-storage tolerance (3e-2 of the max-norm; tests/test_gpu_bf16_storage.py uses the same)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    from dlwp_benchmark_amd.nsbench import swin_transformer as st
    from dlwp_benchmark_amd.train_engine import flatten_parameters, refresh_bf16_weights
    from oracle import pangu_ref
    B, res, heads, dim = 8, (1, 30, 60), 6, 192
    torch.manual_seed(11)
    L.set_gemm_precision("bf16")
    L.set_storage("bf16")
    try:
        blk = EarthSpecificBlock(dim, res, heads, (2, 7, 7), None if shifted else (0, 0, 0)).to(cuda)
        with torch.no_grad():
            blk.attn.earth_position_bias_table.normal_(0, 0.5)
            blk.attn.qkv.bias.normal_(0, 0.5)
        blk.real_token_flow = True
        flatten_parameters(blk)                    # bf16 weight copies + gradient slots, as train_engine.GraphedTrainStep does
        L_ = res[0] * res[1] * res[2]
        x = torch.randn(B, L_, dim, device=cuda, requires_grad=True)
        g = torch.randn(B, L_, dim, device=cuda)
        assert st._WindowAttnTokensFn.wants_bf16_qkv(B, blk._wspec, heads, dim // heads) is False      # not inside a step yet
        refresh_bf16_weights(blk)
        prev, L.SHADOW_ACTIVE = L.SHADOW_ACTIVE, True
        try:
            assert st._WindowAttnTokensFn.wants_bf16_qkv(B, blk._wspec, heads, dim // heads)
            y = blk(x)
            y.backward(g)
        finally:
            L.SHADOW_ACTIVE = prev
        names = [n for n, _ in blk.named_parameters()]
        got = [x.grad] + [p.grad for p in blk.parameters()]
    finally:
        L.set_storage("fp32")
        L.set_gemm_precision("fp32")
    p64 = {n: t.detach().double().cpu().requires_grad_(True) for n, t in blk.named_parameters()}
    x64 = x.detach().double().cpu().requires_grad_(True)
    y64 = pangu_ref.earth_block(x64, p64, "", res, heads, (2, 7, 7), pangu_ref.DEFAULT_SHIFT if shifted else (0, 0, 0))
    g64 = torch.autograd.grad(y64, [x64] + [p64[n] for n in names], g.double().cpu())
    assert rel(y, y64) <= 3e-2
    for n, a, b in zip(["x"] + names, got, g64):
        assert a is not None, n
        assert rel(a, b) <= 3e-2, n


@pytest.mark.parametrize("N,heads,d,TB,qr", [(49, 4, 32, 169, None), (49, 3, 24, 169, None), (64, 2, 16, 225, (16, 64))])
def test_one_pass_backward_for_short_windows_equals_the_two_pass_kernel(cuda, N, heads, d, TB, qr):
    """Windows of at most 64 tokens take the two-pass LDS kernel by default; the one-pass kernel's 4-wave instantiation (tuning knob
    WINATTN_BWD1P_SMALL) must give the same gradients (bf16 operands on both sides: 2e-3 of the max-norm; float atomics reorder the
    bias-table sums)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_core
    torch.manual_seed(3)
    nW, B_ = 8, 1024
    qkv = torch.randn(B_, N, 3 * heads * d, device=cuda)
    table = (torch.randn(TB, heads, device=cuda) * 0.5)
    ia = torch.randint(0, TB // 2, (N,), device=cuda, dtype=torch.int32)
    ib = torch.randint(0, TB // 2, (N,), device=cuda, dtype=torch.int32)
    labels = torch.randint(0, 3, (nW, N), device=cuda, dtype=torch.int32)
    g = torch.randn(B_, N, heads * d, device=cuda)
    if qr is not None:
        g[:, :qr[0]] = 0
    res = {}
    with L.gemm_precision("bf16"):
        for small in (0, 1):
            L.set_tuning("WINATTN_BWD1P_SMALL", small)
            try:
                q_, t_ = qkv.clone().requires_grad_(True), table.clone().requires_grad_(True)
                y = window_attention_core(q_, t_, ia, ib, labels, nW, heads, d ** -0.5, qr)
                res[small] = torch.autograd.grad(y, (q_, t_), g)
            finally:
                L.set_tuning("WINATTN_BWD1P_SMALL", None)
    for a, b in zip(res[1], res[0]):
        assert rel(a, b) <= 2e-3


@pytest.mark.parametrize("shifted", [False, True])
@pytest.mark.parametrize("io", ["fp32", "bf16"])
def test_lds_staged_token_forward_equals_the_wave_per_window_forward(cuda, shifted, io):
    """dlwp_window_attn_fwd_tokens has two kernels (tuning knob WINATTN_FWD_LDS): same operands, same rounding points -> outputs,
    row statistics (through the backward) and gradients agree to 2e-3 of the max-norm (bf16 products, different summation order)."""
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    from dlwp_benchmark_amd.nsbench.swin_transformer import window_attention_tokens
    torch.manual_seed(13)
    B, res, heads, dim = 8, (1, 30, 60), 6, 192
    blk = EarthSpecificBlock(dim, res, heads, (2, 7, 7), None if shifted else (0, 0, 0)).to(cuda)
    spec = blk._wspec
    sh = blk.shift_size
    fwd_shift = (sh[0], sh[1], sh[1]) if blk.roll else (0, 0, 0)
    rev_shift = sh if blk.roll else (0, 0, 0)
    Ltok = res[0] * res[1] * res[2]
    qkv = torch.randn(B, Ltok, 3 * dim, device=cuda)
    fill = torch.randn(3 * dim, device=cuda) * 0.5
    table = torch.randn_like(blk.attn.earth_position_bias_table) * 0.5
    g = torch.randn(B, Ltok, dim, device=cuda)
    if io == "bf16":
        qkv, g = qkv.to(torch.bfloat16), g.to(torch.bfloat16)
    res_ = {}
    with L.gemm_precision("bf16"):
        for knob in (0, 1):
            L.set_tuning("WINATTN_FWD_LDS", knob)
            try:
                q_, f_, t_ = qkv.clone().requires_grad_(True), fill.clone().requires_grad_(True), table.clone().requires_grad_(True)
                y = window_attention_tokens(q_, f_, t_, blk.attn._ia, blk.attn._ib, blk._labels if blk.roll else None, spec, fwd_shift,
                                            rev_shift, heads, float(blk.attn.scale), blk._qrange)
                res_[knob] = (y.detach().float(),) + tuple(t.float() for t in torch.autograd.grad(y, (q_, f_, t_), g))
            finally:
                L.set_tuning("WINATTN_FWD_LDS", None)
    for a, b in zip(res_[1], res_[0]):
        assert torch.isfinite(a).all()
        assert rel(a, b) <= (1e-2 if io == "bf16" else 2e-3)
