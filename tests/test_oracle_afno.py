"""The AFNO oracle (oracle/afno_ref.py) against golden vectors captured from the reference's own
classes (tests/golden/make_afno_golden.py).  This is what pins the oracle: bit-level agreement is not
expected (complex einsum vs four real einsums), 1e-5 relative is."""
import os

import numpy as np
import pytest
import torch

from oracle import afno_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "afno_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tag", ["sq", "rect", "frac"])
def test_afno2d_matches_reference(tag):
    B, H, W, C, nb, frac100 = [int(v) for v in G[f"afno2d_{tag}_meta"]]
    x = t(f"afno2d_{tag}_x").requires_grad_(True)
    ps = [t(f"afno2d_{tag}_{n}").requires_grad_(True) for n in ("w1", "b1", "w2", "b2")]
    y = afno_ref.afno2d(x, *ps, nb, 0.01, frac100 / 100.0)
    assert rel(y.detach(), t(f"afno2d_{tag}_y")) < 1e-5
    y.backward(t(f"afno2d_{tag}_gy"))
    assert rel(x.grad, t(f"afno2d_{tag}_gx")) < 1e-5
    for p, n in zip(ps, ("gw1", "gb1", "gw2", "gb2")):
        assert rel(p.grad, t(f"afno2d_{tag}_{n}")) < 2e-5, n


def test_kept_mode_quirk_32x64():
    # on 32x64 only 17 of the 33 rfft columns are mixed, the rest are zeroed
    assert afno_ref.kept_window(32, 64, 1.0) == (0, 32, 17)
    assert afno_ref.kept_window(16, 16, 1.0) == (0, 16, 9)
    assert afno_ref.kept_window(16, 32, 0.5) == (5, 13, 4)


def test_block_matches_reference():
    p = {k[len("block_p_"):]: t(k).requires_grad_(True) for k in G.files if k.startswith("block_p_")}
    x = t("block_x").requires_grad_(True)
    y = afno_ref.block(x, p, "", 4)
    assert rel(y.detach(), t("block_y")) < 1e-5
    y.backward(t("block_gy"))
    assert rel(x.grad, t("block_gx")) < 1e-5
    for n, v in p.items():
        assert rel(v.grad, t("block_g_" + n)) < 5e-5, n


def test_afnonet_rollout_matches_reference():
    p = {k[len("net_p_"):]: t(k).requires_grad_(True) for k in G.files if k.startswith("net_p_")}
    cfg = dict(img_height=32, img_width=32, patch_size=(4, 4), in_chans=1, out_chans=1, embed_dim=32, depth=2,
               num_blocks=4, context_size=2)
    y = afno_ref.afnonet(t("net_x"), p, cfg, teacher_forcing_steps=3)
    assert rel(y.detach(), t("net_y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, t("net_target"))
    assert abs(loss.item() - float(G["net_loss"])) < 1e-6 * abs(float(G["net_loss"])) + 1e-7
    loss.backward()
    for n, v in p.items():
        if "net_g_" + n in G.files:
            assert rel(v.grad, t("net_g_" + n)) < 1e-4, n
        else:
            assert v.grad is None  # AFNONet.norm is never used by the reference forward


# ---- dlwpbench twin --------------------------------------------------------------------------------------
GD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dlwp_afno_golden.npz"))
DLWP_CFG = {"one": dict(img_height=16, img_width=32, patch_size=(2, 2), prognostic_channels=3, embed_dim=32, depth=2,
                        num_blocks=4, context_size=1),
            "multi": dict(img_height=16, img_width=32, patch_size=(4, 4), prognostic_channels=2, embed_dim=32, depth=2,
                          num_blocks=4, context_size=2)}


@pytest.mark.parametrize("tag", ["one", "multi"])
def test_dlwp_afnonet_matches_reference(tag):
    """`one`: the reference's own forward(); `multi`: the reference's layers driven by the clean loop."""
    td = lambda n: torch.from_numpy(GD[f"{tag}_{n}"])   # noqa: E731
    p = {k[len(tag) + 3:]: torch.from_numpy(GD[k]).clone().requires_grad_(True) for k in GD.files if k.startswith(f"{tag}_p_")}
    y = afno_ref.dlwp_afnonet(td("constants"), td("prescribed"), td("prognostic"), p, DLWP_CFG[tag])
    assert rel(y.detach(), td("y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, td("target"))
    assert abs(loss.item() - float(GD[f"{tag}_loss"])) < 1e-5 * abs(float(GD[f"{tag}_loss"]))
    loss.backward()
    for n, v in p.items():
        if f"{tag}_g_{n}" in GD.files:
            assert rel(v.grad, td(f"g_{n}")) < 1e-4, n
