"""GPU parity of the Pangu-Weather path against golden vectors captured from the reference's own classes
(tests/golden/pangu_golden.npz).  Tolerance: 1e-4 forward, 5e-4..2e-3 gradients (fp32, max-norm)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pangu_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def load(module, prefix):
    sd = {k[len(prefix):]: t(k) for k in G.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(("earth_position_index" in m) or ("attn_mask" in m) for m in missing), missing


@pytest.mark.parametrize("tag,shift", [("plain", (0, 0, 0)), ("shift", None)])
def test_earth_specific_block_matches_reference_golden(cuda, tag, shift):
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    blk = EarthSpecificBlock(dim=16, input_resolution=(1, 10, 20), num_heads=2, window_size=(2, 6, 12), shift_size=shift)
    load(blk, f"blk_{tag}_p_")
    blk = blk.to(cuda).eval()
    x = t(f"blk_{tag}_x").to(cuda).requires_grad_(True)
    y = blk(x)
    assert rel(y, t(f"blk_{tag}_y")) <= 1e-4
    y.backward(t(f"blk_{tag}_gy").to(cuda))
    assert rel(x.grad, t(f"blk_{tag}_gx")) <= 5e-4
    for n, p in blk.named_parameters():
        assert rel(p.grad, t(f"blk_{tag}_g_{n}")) <= 1e-3, n


def test_pangu_rollout_step_matches_reference_golden(cuda):
    from dlwp_benchmark_amd import dlwpbench
    net = dlwpbench.PanguWeather(constant_channels=2, prescribed_channels=1, prognostic_channels=3, embed_dim=8,
                                 num_heads=(1, 2, 2, 1), window_size=(2, 4, 8), patch_size=(1, 1), n_lat=18, n_lon=32,
                                 context_size=1, type="PanguWeather", name="t")
    load(net, "net_p_")
    net = net.to(cuda).eval()
    y = net(constants=t("net_constants").to(cuda), prescribed=t("net_prescribed").to(cuda),
            prognostic=t("net_prognostic").to(cuda))
    assert rel(y, t("net_y")) <= 1e-4
    loss = torch.nn.functional.mse_loss(y, t("net_target").to(cuda))
    assert abs(loss.item() - float(G["net_loss"])) <= 1e-4 * abs(float(G["net_loss"]))
    loss.backward()
    for n, p in net.named_parameters():
        if "net_g_" + n in G.files:
            assert rel(p.grad, t("net_g_" + n)) <= 3e-3, n
