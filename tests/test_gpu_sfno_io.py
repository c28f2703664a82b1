"""GPU parity of the one-launch SFNO encoder / decoder (csrc/sfno_io.hip, dlwp_benchmark_amd/sfno_ops.py) against
(i) a float64 restatement of the encoder / decoder MLPs with the kernels' rounding points (operands and stored hidden tensors
rounded to bf16; network: torch_harmonics' SFNO as constructed at /root/reference/src/dlwpbench/models/fno/fno.py:183-200,
SURVEY.md App. A-2; frame bookkeeping: fno.py:217-259 / unet.py:64-111) and (ii) the GEMM path of dlwpbench/sfno.py that it
replaces, on whole SFNO2DModule rollouts (outputs and every parameter gradient, bf16 operands + bf16 storage in both).
Tolerances: 1e-2 relative max-norm on fp32 results computed from bf16-rounded hidden tensors, exact on pure copies."""
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


class _Mode:
    """bf16 operands + bf16 storage with a `current` weight shadow, as inside a train_engine step"""

    def __enter__(self):
        from dlwp_benchmark_amd import lib as L
        self.L = L
        self.p = L.gemm_precision("bf16")
        self.p.__enter__()
        L.set_storage("bf16")
        L.SHADOW_ACTIVE = True
        return self

    def __exit__(self, *exc):
        self.L.SHADOW_ACTIVE = False
        self.L.set_storage("fp32")
        self.p.__exit__(*exc)
        return False


@pytest.mark.parametrize("B,H,W,E,big_skip,pos", [(2, 32, 64, 256, True, True), (1, 16, 32, 64, False, False), (3, 8, 16, 128, True, False)])
def test_network_on_plane_groups_matches_float64_encoder_decoder(cuda, B, H, W, E, big_skip, pos):
    """SFNO with zero blocks (encoder -> decoder): forward_frames over three plane groups with the residual frame against the
    float64 formula; gradients of the frame and of every parameter."""
    from dlwp_benchmark_amd.dlwpbench.sfno import SFNO
    from dlwp_benchmark_amd import sht
    g = torch.Generator().manual_seed(31)
    cc, cp, cg = 4, 1, 5
    cin = cc + cp + cg
    with _Mode():
        torch.manual_seed(5)
        net = SFNO(img_size=(H, W), grid="equiangular", scale_factor=1, in_chans=cin, out_chans=cg, embed_dim=E, num_layers=0,
                   big_skip=big_skip, pos_embed=pos, use_mlp=True, normalization_layer="none").to(cuda)
        if pos:
            with torch.no_grad():
                net.pos_embed.copy_(0.5 * torch.randn(net.pos_embed.shape, generator=g))
        assert net.fast_io(cin)
        const = torch.randn(B, cc, H, W, generator=g)
        presc = torch.randn(B, 3, cp, H, W, generator=g)                  # a [B, T, c, H, W] tensor: frames are strided views
        frame = torch.randn(B, cg, H, W, generator=g)
        gout = torch.randn(B, cg, H, W, generator=g)
        fr = frame.clone().to(cuda).requires_grad_(True)
        with sht.spectral_weight_scope():
            out = net.forward_frames([const.to(cuda), presc.to(cuda)[:, 1], fr], 2, residual=True)
            out.backward(gout.to(cuda))
        torch.cuda.synchronize()
    P = {k: v.detach().double().cpu() for k, v in net.state_dict().items()}
    x = torch.cat([const, presc[:, 1], frame], 1).double()                # [B, cin, H, W]
    xt = x.permute(0, 2, 3, 1).reshape(-1, cin)
    w1e, b1e, w2e = P["encoder.0.weight"].reshape(E, cin), P["encoder.0.bias"], P["encoder.2.weight"].reshape(E, E)
    wd, bd, w2d = P["decoder.0.weight"].reshape(E, -1), P["decoder.0.bias"], P["decoder.2.weight"].reshape(cg, E)
    ze = rb(xt) @ rb(w1e).t() + b1e
    he = rb(gelu64(ze))
    t0 = he @ rb(w2e).t()
    if pos:
        t0 = t0 + P["pos_embed"][0].permute(1, 2, 0).reshape(1, H * W, E).expand(B, -1, -1).reshape(-1, E)
    dec_in = torch.cat([rb(t0), rb(xt)], 1) if big_skip else rb(t0)
    zd = dec_in @ rb(wd).t() + bd
    hd = rb(gelu64(zd))
    y = hd @ rb(w2d).t()
    outr = frame.double() + y.reshape(B, H, W, cg).permute(0, 3, 1, 2)
    assert rel(out, outr) <= 1e-2
    # backward in float64 from the float64 forward (rounding of the stored pre-activations is below the tolerance)
    go = gout.double().permute(0, 2, 3, 1).reshape(-1, cg)
    ghd = rb((rb(go) @ rb(w2d)) * gelu_grad64(rb(zd)))
    gdin = ghd @ rb(wd)
    gt, gtok_d = gdin[:, :E], (gdin[:, E:] if big_skip else torch.zeros(xt.shape[0], cin, dtype=torch.float64))
    ghe = rb((rb(gt) @ rb(w2e)) * gelu_grad64(rb(ze)))
    gx = ghe @ rb(w1e) + gtok_d
    gframe = gout.double() + gx[:, cc + cp:].reshape(B, H, W, cg).permute(0, 3, 1, 2)
    assert rel(fr.grad, gframe) <= 1e-2
    sd = dict(net.named_parameters())
    want = {"decoder.2.weight": rb(go).t() @ hd, "decoder.0.weight": ghd.t() @ dec_in, "decoder.0.bias": ghd.sum(0),
            "encoder.2.weight": rb(gt).t() @ he, "encoder.0.weight": ghe.t() @ rb(xt), "encoder.0.bias": ghe.sum(0)}
    if pos:
        want["pos_embed"] = gt.reshape(B, H, W, E).sum(0).permute(2, 0, 1)
    for k, v in want.items():
        assert sd[k].grad is not None, k
        assert rel(sd[k].grad.reshape(v.shape), v) <= 1e-2, (k, rel(sd[k].grad.reshape(v.shape), v))


@pytest.mark.parametrize("over", [dict(), dict(big_skip=False, pos_embed=False, embed_dim=64, prognostic_channels=8)])
def test_fast_rollout_equals_the_gemm_path_rollout(cuda, over, monkeypatch):
    """SFNO2DModule.forward at context_size 1: the plane-group rollout (one-launch encoder / decoder) against the GEMM-path
    rollout of the same module (rollout.py + cat / pad / permute), outputs and every parameter gradient."""
    from dlwp_benchmark_amd import dlwpbench
    from dlwp_benchmark_amd.dlwpbench import sfno as sfno_mod
    g = torch.Generator().manual_seed(32)
    cfg = dict(constant_channels=4, prescribed_channels=1, prognostic_channels=5, grid="equiangular", num_layers=2, scale_factor=1,
               embed_dim=128, context_size=1, height=32, width=64, big_skip=True, pos_embed=True, use_mlp=True, normalization_layer="none")
    cfg.update(over)
    B, T, cg = 2, 4, cfg["prognostic_channels"]
    data = dict(constants=torch.randn(B, 1, 4, 32, 64, generator=g).to(cuda), prescribed=torch.randn(B, T, 1, 32, 64, generator=g).to(cuda),
                prognostic=torch.randn(B, T, cg, 32, 64, generator=g).to(cuda))
    gout = torch.randn(B, T - 1, cg, 32, 64, generator=g).to(cuda)
    res = {}
    with _Mode():
        for fast in (True, False):
            monkeypatch.setattr(sfno_mod, "FAST_IO", fast)
            torch.manual_seed(9)
            m = dlwpbench.SFNO2DModule(**cfg).to(cuda)
            if m.sfno.pos_embed is not None:
                with torch.no_grad():
                    m.sfno.pos_embed.normal_(0.0, 0.3)
            assert m.sfno.fast_io(m.sfno.encoder[0].in_channels) == fast
            out = m(**data)
            out.backward(gout)
            res[fast] = (out, {k: p.grad for k, p in m.named_parameters()})
    assert rel(res[True][0], res[False][0]) <= 2e-2
    for k in res[False][1]:
        assert res[True][1][k] is not None, k
        assert rel(res[True][1][k], res[False][1][k]) <= 3e-2, (k, rel(res[True][1][k], res[False][1][k]))
