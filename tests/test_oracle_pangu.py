"""The Pangu oracle against golden vectors captured from the reference's own classes."""
import os

import numpy as np
import pytest
import torch

from oracle import pangu_ref

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "pangu_golden.npz"))


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def pdict(prefix):
    return {k[len(prefix):]: t(k).clone().requires_grad_(True) for k in G.files if k.startswith(prefix)}


def test_earth_index_is_additive_in_query_and_key():
    idx = pangu_ref.earth_index((2, 6, 12))
    ia, ib = idx[:, 0] - idx[0, 0], idx[0, :]
    assert torch.equal(ia[:, None] + ib[None, :], idx)          # what the HIP kernel relies on
    assert idx.min() == 0 and idx.max() == 4 * 36 * 23 - 1


@pytest.mark.parametrize("tag,shift", [("plain", (0, 0, 0)), ("shift", (1, 3, 6))])
def test_earth_block_matches_reference(tag, shift):
    p = pdict(f"blk_{tag}_p_")
    x = t(f"blk_{tag}_x").requires_grad_(True)
    y = pangu_ref.earth_block(x, p, "", (1, 10, 20), 2, (2, 6, 12), shift)
    assert rel(y.detach(), t(f"blk_{tag}_y")) < 1e-5
    y.backward(t(f"blk_{tag}_gy"))
    assert rel(x.grad, t(f"blk_{tag}_gx")) < 2e-5
    for n, v in p.items():
        assert rel(v.grad, t(f"blk_{tag}_g_{n}")) < 5e-5, n


def test_pangu_one_step_rollout_matches_reference():
    p = pdict("net_p_")
    cfg = dict(embed_dim=8, num_heads=(1, 2, 2, 1), window_size=(2, 4, 8), patch_size=(1, 1), n_lat=18, n_lon=32,
               context_size=1)
    y = pangu_ref.rollout(t("net_constants"), t("net_prescribed"), t("net_prognostic"), p, cfg)
    assert rel(y.detach(), t("net_y")) < 1e-5
    loss = torch.nn.functional.mse_loss(y, t("net_target"))
    assert abs(loss.item() - float(G["net_loss"])) < 1e-5 * abs(float(G["net_loss"]))
    loss.backward()
    for n, v in p.items():
        if "net_g_" + n in G.files:
            assert rel(v.grad, t("net_g_" + n)) < 5e-4, n
