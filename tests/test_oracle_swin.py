"""The Swin / window-attention oracle against golden vectors captured from the reference's own classes."""
import os

import numpy as np
import pytest
import torch

from oracle import swin_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "swin_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def pdict(prefix):
    return {k[len(prefix):]: t(k).clone().requires_grad_(True) for k in G.files if k.startswith(prefix)}


@pytest.mark.parametrize("tag", ["nomask", "mask"])
def test_window_attention_matches_reference(tag):
    p = pdict("wa_p_")
    x = t(f"wa_{tag}_x").requires_grad_(True)
    labels = torch.from_numpy(G["wa_labels"]) if tag == "mask" else None
    y = swin_ref.window_attention(x, p, "", 7, 7, 2, labels)
    assert rel(y.detach(), t(f"wa_{tag}_y")) < 1e-5
    y.backward(t(f"wa_{tag}_gy"))
    assert rel(x.grad, t(f"wa_{tag}_gx")) < 1e-5
    for n, v in p.items():
        assert rel(v.grad, t(f"wa_{tag}_g_{n}")) < 2e-5, n


@pytest.mark.parametrize("tag,H,W,pm", [("28x28", 28, 28, "constant"), ("20x30", 20, 30, "circular")])
def test_basic_layer_window7_matches_reference(tag, H, W, pm):
    p = pdict(f"bl_{tag}_p_")
    x = t(f"bl_{tag}_x").requires_grad_(True)
    y = swin_ref.basic_layer(x, p, "", H, W, 7, 2, 2, False, pm)[0]
    assert rel(y.detach(), t(f"bl_{tag}_y")) < 1e-5
    y.backward(t(f"bl_{tag}_gy"))
    assert rel(x.grad, t(f"bl_{tag}_gx")) < 2e-5
    for n, v in p.items():
        assert rel(v.grad, t(f"bl_{tag}_g_{n}")) < 5e-5, n


def test_swin_rollout_matches_reference():
    p = pdict("net_p_")
    cfg = dict(context_size=2, pretrain_img_size=32, patch_size=2, embed_dim=8, depths=[2, 2], num_heads=[2, 2])
    y = swin_ref.swin_rollout(t("net_x"), p, cfg, teacher_forcing_steps=2)
    assert rel(y.detach(), t("net_y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, t("net_target"))
    assert abs(loss.item() - float(G["net_loss"])) < 1e-5 * abs(float(G["net_loss"]))
    loss.backward()
    for n, v in p.items():
        if "net_g_" + n in G.files:
            assert rel(v.grad, t("net_g_" + n)) < 2e-4, n


# ---- dlwpbench twin --------------------------------------------------------------------------------------
GD = np.load(os.path.join(os.path.dirname(__file__), "golden", "dlwp_swin_golden.npz"))
DLWP_CFG = {"one": dict(context_size=1, img_height=16, img_width=32, patch_size=2, embed_dim=8, depths=[2, 2], num_heads=[2, 2]),
            "multi": dict(context_size=2, img_height=16, img_width=32, patch_size=1, embed_dim=8, depths=[2, 2], num_heads=[2, 2])}


@pytest.mark.parametrize("tag", ["one", "multi"])
def test_dlwp_swin_matches_reference(tag):
    """`one`: the reference's own forward(); `multi`: the reference's one_step() driven by the clean loop."""
    td = lambda n: torch.from_numpy(GD[f"{tag}_{n}"])   # noqa: E731
    p = {k[len(tag) + 3:]: torch.from_numpy(GD[k]).clone().requires_grad_(True) for k in GD.files if k.startswith(f"{tag}_p_")}
    y = swin_ref.dlwp_swin(td("constants"), td("prescribed"), td("prognostic"), p, DLWP_CFG[tag])
    assert rel(y.detach(), td("y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, td("target"))
    assert abs(loss.item() - float(GD[f"{tag}_loss"])) < 1e-5 * abs(float(GD[f"{tag}_loss"]))
    loss.backward()
    for n, v in p.items():
        if f"{tag}_g_{n}" in GD.files:
            assert rel(v.grad, td(f"g_{n}")) < 2e-4, n


# ---- ape=True (absolute position embedding) ----------------------------------------------------------------
GA = np.load(os.path.join(os.path.dirname(__file__), "golden", "swin_ape_golden.npz"))
NS_APE_CFG = dict(context_size=2, pretrain_img_size=32, patch_size=2, embed_dim=8, depths=[2, 2], num_heads=[2, 2])


def _ape_case(tag):
    td = lambda n: torch.from_numpy(GA[f"{tag}_{n}"])   # noqa: E731
    p = {k[len(tag) + 3:]: torch.from_numpy(GA[k]).clone().requires_grad_(True) for k in GA.files if k.startswith(f"{tag}_p_")}
    assert "absolute_pos_embed" in p
    return td, p


@pytest.mark.parametrize("tag", ["ns", "ns_resized"])
def test_ns_swin_ape_matches_reference(tag):
    """`ns`: frame = pretraining size (the bicubic resize is the identity); `ns_resized`: a 24 x 24 frame on the 16 x 16 embedding."""
    td, p = _ape_case(tag)
    y = swin_ref.swin_rollout(td("x"), p, NS_APE_CFG, teacher_forcing_steps=2)
    assert rel(y.detach(), td("y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, td("target"))
    loss.backward()
    for n, v in p.items():
        assert rel(v.grad, td(f"g_{n}")) < 2e-4, n


def test_dlwp_swin_ape_matches_reference():
    td, p = _ape_case("dlwp")
    y = swin_ref.dlwp_swin(td("constants"), td("prescribed"), td("prognostic"), p, DLWP_CFG["multi"])
    assert rel(y.detach(), td("y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, td("target"))
    loss.backward()
    for n, v in p.items():
        assert rel(v.grad, td(f"g_{n}")) < 2e-4, n
