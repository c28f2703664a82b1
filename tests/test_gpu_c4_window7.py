"""GPU parity of the window-7 attention layers at BASELINE config C4's grid against golden vectors captured from the
reference's own classes (tests/golden/c4_window7_golden.npz, make_c4_window7_golden.py): nsbench BasicLayer(window 7) on
32 x 64 and 128 x 256 (constant and circular padding to multiples of 7) and a shifted Pangu EarthSpecificBlock with
window (2, 7, 7) at (1, 128, 256).  The big tensors are stored every STRIDE-th token; inputs come from the recorded seeds.
Tolerance: 1e-4 forward, 5e-4 input gradients, 2e-3 parameter gradients (sums over 32768 tokens), fp32 max-norm."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4_window7_golden.npz"))
STRIDE = int(G["stride"])


def t(name):
    return torch.from_numpy(G[name])


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def seeded(seed, *shape):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def load(module, prefix, allow_missing=()):
    sd = {k[len(prefix):]: t(k) for k in G.files if k.startswith(prefix)}
    missing, unexpected = module.load_state_dict(sd, strict=False)
    assert not unexpected, unexpected
    assert all(any(a in m for a in allow_missing) for m in missing), missing


@pytest.mark.parametrize("tag", ["32x64", "128x256", "128x256c"])
def test_basic_layer_window7_at_c4_grids_matches_reference_golden(cuda, tag):
    from dlwp_benchmark_amd.nsbench.swin_transformer import BasicLayer
    H, W, B, seed, circ = [int(v) for v in G[f"bl_{tag}_cfg"]]
    bl = BasicLayer(dim=16, depth=2, num_heads=4, window_size=7, padding_mode="circular" if circ else "constant")
    load(bl, f"bl_{tag}_p_", allow_missing=("relative_position_index",))
    bl = bl.to(cuda)
    x0 = seeded(seed, B, H * W, 16)
    assert abs(x0.double().abs().sum().item() - float(G[f"bl_{tag}_xsum"])) < 1e-6 * float(G[f"bl_{tag}_xsum"]), "RNG stream differs"
    x = x0.to(cuda).requires_grad_(True)
    y = bl(x, H, W)[0]
    assert rel(y[:, ::STRIDE], t(f"bl_{tag}_y")) <= 1e-4
    y.backward(seeded(seed + 100, B, H * W, 16).to(cuda))
    assert rel(x.grad[:, ::STRIDE], t(f"bl_{tag}_gx")) <= 5e-4
    for n, p in bl.named_parameters():
        assert rel(p.grad, t(f"bl_{tag}_g_{n}")) <= 2e-3, n


def test_pangu_block_window_2_7_7_at_128x256_matches_reference_golden(cuda):
    from dlwp_benchmark_amd.dlwpbench.panguweather import EarthSpecificBlock
    blk = EarthSpecificBlock(dim=8, input_resolution=(1, 128, 256), num_heads=2, window_size=(2, 7, 7), shift_size=None)
    load(blk, "pg_p_", allow_missing=("earth_position_index", "attn_mask"))
    blk = blk.to(cuda).eval()
    x0 = seeded(21, 1, 128 * 256, 8)
    assert abs(x0.double().abs().sum().item() - float(G["pg_xsum"])) < 1e-6 * float(G["pg_xsum"]), "RNG stream differs"
    x = x0.to(cuda).requires_grad_(True)
    y = blk(x)
    assert rel(y[:, ::STRIDE], t("pg_y")) <= 1e-4
    y.backward(seeded(121, 1, 128 * 256, 8).to(cuda))
    assert rel(x.grad[:, ::STRIDE], t("pg_gx")) <= 5e-4
    for n, p in blk.named_parameters():
        assert rel(p.grad, t(f"pg_g_{n}")) <= 2e-3, n
