"""GPU parity of the spherical path (batched-GEMM SHT, per-degree complex weights, SFNO network and rollout) against
the CPU oracle oracle/sfno_ref.py (torch rfft + einsum restatement of torch-harmonics; PARITY UNPINNED, see its
header).  Tolerance: 1e-4 relative (max-norm) forward, 1e-3 gradients (fp32)."""
import pytest
import torch

from oracle import sfno_ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def spec_to_complex(X):
    """[L, B, M, 2, C] -> complex [B, C, L, M]"""
    return torch.complex(X[..., 0, :], X[..., 1, :]).permute(1, 3, 0, 2)


@pytest.mark.parametrize("nlat,nlon,lmax,grid,B,C", [(32, 64, 32, "equiangular", 2, 8), (32, 64, 32, "legendre-gauss", 1, 12),
                                                      (16, 32, 11, "legendre-gauss", 3, 5), (19, 40, 19, "equiangular", 2, 7),
                                                      # channel counts / order counts the fused single-launch kernels take
                                                      (32, 64, 32, "equiangular", 2, 16), (32, 64, 32, "legendre-gauss", 1, 48),
                                                      (16, 32, 16, "legendre-gauss", 3, 32), (24, 48, 16, "equiangular", 2, 16)])
def test_sht_and_inverse_match_oracle(cuda, nlat, nlon, lmax, grid, B, C):
    from dlwp_benchmark_amd import sht
    o = sfno_ref.SHT(nlat, nlon, lmax, lmax, grid)
    fwd = sht.RealSHT(nlat, nlon, lmax, lmax, grid).to(cuda)
    inv = sht.InverseRealSHT(nlat, nlon, lmax, lmax, grid).to(cuda)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, nlat, nlon, generator=g)
    xr = x.clone().requires_grad_(True)
    Xr = o.forward(xr)
    xd = x.permute(0, 2, 3, 1).contiguous().to(cuda).requires_grad_(True)
    X = fwd(xd)
    assert X.shape == (lmax, B, lmax, 2, C)
    assert rel(torch.view_as_real(spec_to_complex(X).contiguous()), torch.view_as_real(Xr.detach())) <= 1e-4
    # inverse and the gradients of the composition
    yr = o.inverse(Xr)
    y = inv(X)
    assert rel(y.permute(0, 3, 1, 2), yr) <= 1e-4
    gy = torch.randn(B, C, nlat, nlon, generator=g)
    yr.backward(gy)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(cuda))
    assert rel(xd.grad.permute(0, 3, 1, 2), xr.grad) <= 1e-3


@pytest.mark.parametrize("nlat,nlon,lmax,grid,B,C", [(32, 64, 32, "equiangular", 4, 256), (24, 48, 16, "legendre-gauss", 3, 32),
                                                      (20, 64, 24, "equiangular", 2, 16)])
def test_fused_sht_kernels_equal_the_gemm_path(cuda, nlat, nlon, lmax, grid, B, C):
    """csrc/sht_fused.hip (one launch per transform, intermediate in LDS) against the two-GEMM path of the same tables:
    forward and both backward passes (each backward is the other fused kernel with transposed tables); ragged latitude
    counts (padding to 16) included."""
    from dlwp_benchmark_amd import lib as L, sht
    assert L.load().dlwp_sht_fused_supported(nlat, nlon, C, lmax, lmax) == 1
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, nlat, nlon, C, generator=g).to(cuda)
    gX = torch.randn(lmax, B, lmax, 2, C, generator=g).to(cuda)
    gx = torch.randn(B, nlat, nlon, C, generator=g).to(cuda)
    res = {}
    for fused in (True, False):
        fwd = sht.RealSHT(nlat, nlon, lmax, lmax, grid, fused=fused).to(cuda)
        inv = sht.InverseRealSHT(nlat, nlon, lmax, lmax, grid, fused=fused).to(cuda)
        xa = x.clone().requires_grad_(True)
        X = fwd(xa)
        X.backward(gX)
        Xb = gX.clone().requires_grad_(True)
        y = inv(Xb)
        y.backward(gx)
        res[fused] = (X.detach(), xa.grad, y.detach(), Xb.grad)
    for a_, b_ in zip(res[True], res[False]):
        assert rel(a_, b_) <= 2e-5



@pytest.mark.parametrize("nlat,nlon,lmax,grid,B,C", [(32, 64, 32, "equiangular", 4, 256), (32, 64, 32, "legendre-gauss", 2, 64),
                                                      (24, 48, 16, "legendre-gauss", 3, 32), (40, 96, 24, "equiangular", 1, 16),
                                                      (64, 128, 32, "legendre-gauss", 1, 48)])
def test_bf16_fused_sht_kernels_equal_the_bf16_gemm_path(cuda, monkeypatch, nlat, nlon, lmax, grid, B, C):
    """csrc/sht_bf16.hip (one launch per transform, bf16 MFMA, intermediate plane in LDS) against the two bf16 table GEMMs of the
    same chain (bf16 operands + bf16 storage in both): analysis, synthesis, both backward passes and the forked-input skip
    gradient; padded latitude / longitude / degree counts included.  Tolerance 1e-2 (bf16 rounding of the intermediate plane may
    fall on either side), and 2e-2 against the float64 transform of the bf16-rounded input."""
    from dlwp_benchmark_amd import lib as L, sht
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, nlat, nlon, C, generator=g).to(cuda)
    gX = torch.randn(lmax, B, lmax, 2, C, generator=g).to(cuda)
    gs = torch.randn(B, nlat, nlon, C, generator=g).to(cuda)
    fwd = sht.RealSHT(nlat, nlon, lmax, lmax, grid).to(cuda)
    inv = sht.InverseRealSHT(nlat, nlon, lmax, lmax, grid).to(cuda)
    res = {}
    with L.gemm_precision("bf16"):
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        try:
            assert L.load().dlwp_sht_bf16_supported(nlat, nlon, C, lmax, lmax) == 1
            for mode in ("1", "0"):
                monkeypatch.setenv("DLWP_SHT_BF16", mode)
                xa = x.clone().requires_grad_(True)
                X, skip = fwd(xa, fork=True)
                assert X.dtype == torch.bfloat16
                (X.float() * gX).sum().backward(retain_graph=True)
                g_plain = xa.grad.clone()
                xa.grad = None
                ((X.float() * gX).sum() + (skip * gs).sum()).backward()
                Xl = X.detach().clone().requires_grad_(True)
                y = inv(Xl)
                y.backward(gs)
                res[mode] = (X.detach().float(), g_plain, xa.grad.clone(), y.detach(), Xl.grad.float())
        finally:
            L.SHADOW_ACTIVE = False
            L.set_storage("fp32")
    for i, (a, b) in enumerate(zip(res["1"], res["0"])):
        assert rel(a, b) <= 1e-2, (i, rel(a, b))
    assert rel(res["1"][2] - res["1"][1], gs) <= 1e-2          # the skip gradient arrives unchanged on top of the transform's
    # float64 transform of the rounded input
    o = sfno_ref.SHT(nlat, nlon, lmax, lmax, grid)
    Xr = o.forward(x.cpu().to(torch.bfloat16).double().permute(0, 3, 1, 2))
    assert rel(torch.view_as_real(spec_to_complex(res["1"][0]).contiguous()), torch.view_as_real(Xr)) <= 2e-2



@pytest.mark.parametrize("variant", [{}, {"DHCONV_RC": 64}, {"DHCONV_APPLY": 1, "DHCONV_PACK": 1}], ids=["pipelined128", "pipelined64", "round4"])
@pytest.mark.parametrize("B,M,L,Ci,Co,tri", [(4, 32, 32, 256, 256, True), (2, 16, 24, 128, 256, True), (3, 32, 20, 256, 128, False),
                                              (5, 12, 12, 128, 128, True), (9, 32, 32, 128, 128, True), (1, 32, 32, 256, 256, True)])
def test_native_dhconv_kernels_match_the_complex_einsum(cuda, B, M, L, Ci, Co, tri, variant):
    """csrc/dhconv.hip: forward, input gradient and the weight gradient over THREE applications in one scope (one segmented product)
    against the float64 complex einsum "bixy,iox->boxy" on bf16-rounded operands; triangular spectra (orders m > l zero, as RealSHT
    produces them) with the skip on, dense spectra with it off, ragged row counts (partial chunks), one to five chunks per
    workgroup walk; every apply kernel variant (pipelined with 128- / 64-row chunks, the round-4 one-chunk kernel) and both
    weight-image pack kernels (one read of the weight for both images; one read per image)."""
    from dlwp_benchmark_amd import lib as L_, sht
    for k_, v_ in variant.items():
        L_.set_tuning(k_, v_)
    g = torch.Generator().manual_seed(14)
    bf = torch.bfloat16
    w = (torch.randn(Ci, Co, L, 2, generator=g) / Ci ** 0.5)
    mask = (torch.arange(M)[None, :] <= torch.arange(L)[:, None]).float() if tri else torch.ones(L, M)       # [L, M]
    Xs = [(torch.randn(L, B, M, 2, Ci, generator=g) * mask[:, None, :, None, None]).to(bf) for _ in range(3)]
    gYs = [(torch.randn(L, B, M, 2, Co, generator=g) * mask[:, None, :, None, None]).to(bf) for _ in range(3)]
    wc = torch.complex(w[..., 0].to(bf).double(), w[..., 1].to(bf).double())                               # [Ci, Co, L]
    with L_.gemm_precision("bf16"):
        L_.set_storage("bf16")
        L_.SHADOW_ACTIVE = True
        try:
            assert L_.load().dlwp_dhconv_supported(Ci, Co, L) == 1
            wd = w.clone().to(cuda).requires_grad_(True)
            xs = [x.clone().to(cuda).requires_grad_(True) for x in Xs]
            with sht.spectral_weight_scope():
                ys = [sht.dhconv(x, wd, triangular=tri) for x in xs]
                assert all(y.dtype == bf for y in ys)
                sum((y.float() * gy.to(cuda).float()).sum() for y, gy in zip(ys, gYs)).backward()
        finally:
            L_.SHADOW_ACTIVE = False
            L_.set_storage("fp32")
            for k_ in variant:
                L_.set_tuning(k_, None)
    gw_ref = torch.zeros(Ci, Co, L, dtype=torch.complex128)
    for x, gy, y, xd in zip(Xs, gYs, ys, xs):
        xc = torch.complex(x[..., 0, :].double(), x[..., 1, :].double())                                    # [L, B, M, Ci]
        gc = torch.complex(gy[..., 0, :].double(), gy[..., 1, :].double())
        yr = torch.einsum("lbmi,iol->lbmo", xc, wc)
        assert rel(torch.stack([y[..., 0, :], y[..., 1, :]], -1).float(), torch.view_as_real(yr)) <= 1e-2
        gxr = torch.einsum("lbmo,iol->lbmi", gc, wc.conj())
        assert rel(torch.stack([xd.grad[..., 0, :], xd.grad[..., 1, :]], -1).float(), torch.view_as_real(gxr)) <= 1e-2
        gw_ref += torch.einsum("lbmi,lbmo->iol", xc.conj(), gc)
    assert rel(wd.grad, torch.view_as_real(gw_ref)) <= 1e-2


def test_dhconv_matches_einsum(cuda):
    from dlwp_benchmark_amd import sht
    g = torch.Generator().manual_seed(6)
    Lm, B, M, Cin, Cout = 12, 3, 12, 20, 28
    X = torch.randn(Lm, B, M, 2, Cin, generator=g)
    w = torch.randn(Cin, Cout, Lm, 2, generator=g) * 0.2
    gY = torch.randn(Lm, B, M, 2, Cout, generator=g)
    Xr, wr = X.clone().requires_grad_(True), w.clone().requires_grad_(True)
    Yr = torch.einsum("bixy,iox->boxy", spec_to_complex(Xr), torch.view_as_complex(wr))
    Yr.backward(spec_to_complex(gY))
    Xd, wd = X.to(cuda).requires_grad_(True), w.to(cuda).requires_grad_(True)
    Y = sht.dhconv(Xd, wd)
    Y.backward(gY.to(cuda))
    assert rel(torch.view_as_real(spec_to_complex(Y).contiguous()), torch.view_as_real(Yr.detach())) <= 1e-4
    assert rel(Xd.grad, Xr.grad) <= 1e-3
    assert rel(wd.grad, wr.grad) <= 1e-3


CFG = dict(constant_channels=2, prescribed_channels=1, prognostic_channels=3, grid="equiangular", num_layers=3, scale_factor=1,
           embed_dim=16, context_size=1, height=16, width=32, big_skip=True, pos_embed=True, use_mlp=True,
           normalization_layer="none")


# 6 / 10 input channels with embed_dim 16: the frame rows and first encoder / decoder weights run zero-padded to 8 / 16
# channels (sfno.py forward); embed_dim 12 takes the unpadded path
@pytest.mark.parametrize("over", [dict(), dict(context_size=2, scale_factor=2, big_skip=False, pos_embed=False),
                                  dict(embed_dim=12)])
def test_sfno2d_rollout_matches_oracle(cuda, over):
    from dlwp_benchmark_amd import dlwpbench
    cfg = dict(CFG, **over)
    torch.manual_seed(9)
    net = dlwpbench.SFNO2DModule(**cfg)
    with torch.no_grad():
        if net.sfno.pos_embed is not None:
            net.sfno.pos_embed.normal_(0, 0.5)
    g = torch.Generator().manual_seed(10)
    T, ctx, H, W = cfg["context_size"] + 3, cfg["context_size"], cfg["height"], cfg["width"]
    constants = torch.randn(2, 1, 2, H, W, generator=g)
    prescribed = torch.randn(2, T, 1, H, W, generator=g)
    prognostic = torch.randn(2, T, 3, H, W, generator=g)
    target = torch.randn(2, T - ctx, 3, H, W, generator=g)
    p = {k[len("sfno."):]: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    yr = sfno_ref.sfno2d_rollout(constants, prescribed, prognostic, p, cfg)
    torch.nn.functional.mse_loss(yr, target).backward()
    net = net.to(cuda).train()
    y = net(constants=constants.to(cuda), prescribed=prescribed.to(cuda), prognostic=prognostic.to(cuda))
    assert rel(y, yr) <= 1e-4
    torch.nn.functional.mse_loss(y, target.to(cuda)).backward()
    for n, q in net.named_parameters():
        assert rel(q.grad, p[n[len("sfno."):]].grad) <= 2e-3, n


def test_sfno2d_c3_widths_fp32_path_matches_oracle(cuda):
    """The fp32 product path of the BENCHMARKED module -- sfno.yaml widths (embed 256, 4 layers, equiangular 32 x 64, big skip,
    position embedding, MLP), BASELINE configs[2] channel counts, 2 lead times -- against oracle/sfno_ref.py: output 1e-4,
    every parameter gradient 2e-3 (the bf16 path at this width is compared with this fp32 path in test_gpu_fullsize.py)."""
    import bench
    from dlwp_benchmark_amd import dlwpbench
    cfg = dict(bench.SFNO_WORKLOAD["model"])
    torch.manual_seed(31)
    net = dlwpbench.SFNO2DModule(**cfg)
    with torch.no_grad():
        net.sfno.pos_embed.normal_(0, 0.5)
    g = torch.Generator().manual_seed(32)
    B, T, H, W = 2, 3, cfg["height"], cfg["width"]
    constants = torch.randn(B, 1, cfg["constant_channels"], H, W, generator=g)
    prescribed = torch.randn(B, T, cfg["prescribed_channels"], H, W, generator=g)
    prognostic = torch.randn(B, T, cfg["prognostic_channels"], H, W, generator=g)
    target = torch.randn(B, T - 1, cfg["prognostic_channels"], H, W, generator=g)
    p = {k[len("sfno."):]: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    yr = sfno_ref.sfno2d_rollout(constants, prescribed, prognostic, p, cfg)
    torch.nn.functional.mse_loss(yr, target).backward()
    net = net.to(cuda).train()
    y = net(constants=constants.to(cuda), prescribed=prescribed.to(cuda), prognostic=prognostic.to(cuda))
    assert y.shape == yr.shape and rel(y, yr) <= 1e-4, rel(y, yr)
    torch.nn.functional.mse_loss(y, target.to(cuda)).backward()
    for n, q in net.named_parameters():
        assert rel(q.grad, p[n[len("sfno."):]].grad) <= 2e-3, (n, rel(q.grad, p[n[len("sfno."):]].grad))


def test_sfno_graphed_train_step(cuda):
    """Flat parameters + fused gradient accumulation + hipGraph replay follow the eager torch.optim.Adam trajectory."""
    from dlwp_benchmark_amd import dlwpbench
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep, mse_loss
    g = torch.Generator().manual_seed(12)
    kw = dict(constants=torch.randn(2, 1, 2, 16, 32, generator=g).to(cuda), prescribed=torch.randn(2, 4, 1, 16, 32, generator=g).to(cuda),
              prognostic=torch.randn(2, 4, 3, 16, 32, generator=g).to(cuda))
    target = torch.randn(2, 3, 3, 16, 32, generator=g).to(cuda)

    def make():
        torch.manual_seed(13)
        return dlwpbench.SFNO2DModule(**CFG).to(cuda).train()
    ref = make()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    ref_losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = mse_loss(ref(**kw), target)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    step = GraphedTrainStep(make(), kw, target, lr=1e-3, use_graph=True)
    losses = [step().item() for _ in range(4)]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 5e-4 * abs(b), (losses, ref_losses)


def test_sfnonet_fourcastnetv2_matches_oracle_composition(cuda):
    """SFNONet = patch embedding + SFNO + head; checked against the oracle's sfno_net wrapped by hand."""
    import torch.nn.functional as F
    from dlwp_benchmark_amd import dlwpbench
    torch.manual_seed(21)
    cfg = dict(img_height=16, img_width=32, patch_size=(1, 1), constant_channels=2, prescribed_channels=1, prognostic_channels=3,
               grid="equiangular", num_layers=2, scale_factor=1, embed_dim=16, big_skip=True, use_pos_embed=True, use_mlp=True,
               normalization_layer="none", context_size=1)
    net = dlwpbench.FourCastNetv2(**cfg)
    g = torch.Generator().manual_seed(22)
    constants = torch.randn(2, 1, 2, 16, 32, generator=g)
    prescribed = torch.randn(2, 3, 1, 16, 32, generator=g)
    prognostic = torch.randn(2, 3, 3, 16, 32, generator=g)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    p = {k[len("sfno."):]: v for k, v in sd.items() if k.startswith("sfno.")}
    ocfg = dict(height=16, width=32, scale_factor=1, grid="equiangular", num_layers=2, big_skip=True)
    outs = []
    for t in range(1, 3):
        prog_t = prognostic[:, 0:1] if t == 1 else torch.stack(outs, dim=1)[:, -1:]
        x_t = torch.cat([constants[:, 0], prescribed[:, t - 1:t].flatten(1, 2), prog_t.flatten(1, 2)], dim=1)
        e = F.conv2d(x_t, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
        e = (e.flatten(2).transpose(1, 2) + sd["pos_embed"]).reshape(2, 16, 32, 16).permute(0, 3, 1, 2)
        z = sfno_ref.sfno_net(e, p, ocfg).permute(0, 2, 3, 1)
        z = F.linear(z, sd["head.weight"]).permute(0, 3, 1, 2)
        outs.append(prog_t[:, -1] + z)
    yr = torch.stack(outs, dim=1)
    net = net.to(cuda)
    y = net(constants=constants.to(cuda), prescribed=prescribed.to(cuda), prognostic=prognostic.to(cuda))
    assert rel(y, yr) <= 1e-4


def test_sfnonet_shipped_fourcastnetv2_options_match_oracle(cuda):
    """The option set of the shipped fourcastnetv2.yaml (instance_norm, use_mlp False, big_skip False, legendre-gauss data
    grid) against the oracle's sfno_net, forward and every gradient."""
    import torch.nn.functional as F
    from dlwp_benchmark_amd import dlwpbench
    torch.manual_seed(31)
    cfg = dict(img_height=16, img_width=32, patch_size=(1, 1), constant_channels=2, prescribed_channels=1, prognostic_channels=3,
               grid="legendre-gauss", num_layers=3, scale_factor=1, embed_dim=16, big_skip=False, use_pos_embed=True,
               use_mlp=False, normalization_layer="instance_norm", context_size=1)
    net = dlwpbench.FourCastNetv2(**cfg)
    with torch.no_grad():
        for n, q in net.named_parameters():
            if "norm0" in n or "norm1" in n:
                q.add_(0.3 * torch.randn_like(q))           # affine parameters away from (1, 0)
    g = torch.Generator().manual_seed(32)
    constants = torch.randn(2, 1, 2, 16, 32, generator=g)
    prescribed = torch.randn(2, 2, 1, 16, 32, generator=g)
    prognostic = torch.randn(2, 2, 3, 16, 32, generator=g)
    target = torch.randn(2, 1, 3, 16, 32, generator=g)
    sd = {k: v.detach().clone().requires_grad_(v.is_floating_point()) for k, v in net.state_dict().items()}
    p = {k[len("sfno."):]: v for k, v in sd.items() if k.startswith("sfno.")}
    assert "blocks.0.norm0.weight" in p and "blocks.0.mlp.fc1.weight" not in p
    ocfg = dict(height=16, width=32, scale_factor=1, grid="legendre-gauss", num_layers=3, big_skip=False)
    x_t = torch.cat([constants[:, 0], prescribed[:, 0:1].flatten(1, 2), prognostic[:, 0:1].flatten(1, 2)], dim=1)
    e = F.conv2d(x_t, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"])
    e = (e.flatten(2).transpose(1, 2) + sd["pos_embed"]).reshape(2, 16, 32, 16).permute(0, 3, 1, 2)
    z = F.linear(sfno_ref.sfno_net(e, p, ocfg).permute(0, 2, 3, 1), sd["head.weight"]).permute(0, 3, 1, 2)
    yr = (prognostic[:, 0] + z)[:, None]
    F.mse_loss(yr, target).backward()
    net = net.to(cuda).train()
    y = net(constants=constants.to(cuda), prescribed=prescribed.to(cuda), prognostic=prognostic.to(cuda))
    assert rel(y, yr) <= 1e-4
    F.mse_loss(y, target.to(cuda)).backward()
    for n, q in net.named_parameters():
        if q.grad is None or sd[n].grad is None:
            continue
        if sd[n].grad.abs().max() < 1e-8:          # mathematically zero (a per-channel shift in front of the next block's
            assert q.grad.abs().max() < 1e-6, n    # instance norm): both sides hold rounding noise only
            continue
        assert rel(q.grad, sd[n].grad) <= 2e-3, n


def test_sht_bf16_field_flag_is_the_fp32_path_rounded_once(cuda):
    """DLWP_SHT_FIELD_BF16: the synthesis output as a bf16 array equals the fp32 output rounded to bf16 (bit for bit), and the
    analysis of a bf16 field equals the analysis of the same values held in fp32 -- at the C3 shape and at a ragged one."""
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(41)
    for B, K, N, C, M, Lm in ((4, 32, 64, 256, 32, 32), (3, 24, 40, 32, 12, 16)):
        assert lib.dlwp_sht_bf16_supported(K, N, C, M, Lm) == 1
        X = torch.randn(Lm, B, M, 2, C, generator=g).to(cuda).to(bf)
        S1t = (torch.randn(M, K, Lm, generator=g) / 8).to(cuda).to(bf)
        S2 = (torch.randn(N, 2 * M, generator=g) / 8).to(cuda).to(bf)
        y32 = torch.empty(B, K, N, C, device=cuda)
        y16 = torch.empty(B, K, N, C, device=cuda, dtype=bf)
        L.check(lib.dlwp_sht_synthesis_bf16_ex(L.ptr(X), L.ptr(S1t), L.ptr(S2), None, L.ptr(y32), B, K, N, C, M, Lm, 0, L.stream()))
        L.check(lib.dlwp_sht_synthesis_bf16_ex(L.ptr(X), L.ptr(S1t), L.ptr(S2), None, L.ptr(y16), B, K, N, C, M, Lm, 2, L.stream()))
        assert torch.equal(y16, y32.to(bf))
        A1 = (torch.randn(2 * M, N, generator=g) / 8).to(cuda).to(bf)
        A2 = (torch.randn(M, Lm, K, generator=g) / 8).to(cuda).to(bf)
        Xa, Xb = torch.empty_like(X), torch.empty_like(X)
        L.check(lib.dlwp_sht_analysis_bf16_ex(L.ptr(y16.float()), L.ptr(A1), L.ptr(A2), L.ptr(Xa), B, K, N, C, M, Lm, 0, L.stream()))
        L.check(lib.dlwp_sht_analysis_bf16_ex(L.ptr(y16), L.ptr(A1), L.ptr(A2), L.ptr(Xb), B, K, N, C, M, Lm, 2, L.stream()))
        assert torch.equal(Xa, Xb)
    # a bf16 output takes no residual
    assert lib.dlwp_sht_synthesis_bf16_ex(L.ptr(X), L.ptr(S1t), L.ptr(S2), L.ptr(y32), L.ptr(y16), B, K, N, C, M, Lm, 2, L.stream()) != 0
