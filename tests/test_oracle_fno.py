"""CPU checks of the FNO oracle (oracle/fno_ref.py).  neuralop is absent, so the oracle is
"parity unpinned" for the network arithmetic; these tests anchor what can be anchored:
the parameter-count identity against the reference's published budgets, analytic FFT
identities of the spectral layer, and the rollout driver's windowing."""
import torch

from oracle import fno_ref


def test_parameter_budgets_of_the_paper_sweep():
    # src/nsbench/scripts/train_commands.txt:83-91 (hidden widths) <-> plot_results.py:58 budgets
    widths = [2, 8, 27, 38, 54, 77, 108, 154, 217]
    expected = [7067, 50729, 510092, 999119, 2002463, 4051142, 7944029, 16114963, 31947682]
    for w, n in zip(widths, expected):
        net = fno_ref.FNO([12, 12], 10, w, 256, 256, 1, 4)
        assert net.n_params() == n, (w, net.n_params(), n)


def test_spectral_conv_keeps_exactly_the_centred_modes():
    torch.manual_seed(0)
    H, W, m = 64, 64, (12, 12)
    x = torch.randn(2, 3, H, W, dtype=torch.float64)
    w = torch.ones(3, 3, 12, 7, dtype=torch.complex128)
    w = w * torch.eye(3, dtype=torch.complex128)[:, :, None, None]  # identity mixing
    y = fno_ref.spectral_conv(x, w, None, m)
    X = torch.fft.rfft2(x, norm="forward")
    keep = torch.zeros(H, W // 2 + 1, dtype=torch.bool)
    ky = torch.fft.fftfreq(H, 1.0 / H).round().long()  # signed row frequencies
    for r in range(H):
        if -6 <= ky[r] <= 5:
            keep[r, :7] = True
    y_ref = torch.fft.irfft2(X * keep, s=(H, W), norm="forward")
    assert torch.allclose(y, y_ref, atol=1e-12)


def test_spectral_conv_is_linear_and_real():
    torch.manual_seed(1)
    x1 = torch.randn(1, 4, 32, 64, dtype=torch.float64)
    x2 = torch.randn(1, 4, 32, 64, dtype=torch.float64)
    w = torch.randn(4, 5, 8, 5, dtype=torch.complex128)
    f = lambda x: fno_ref.spectral_conv(x, w, None, (8, 9))
    assert f(x1).dtype == torch.float64
    assert torch.allclose(f(2 * x1 - 3 * x2), 2 * f(x1) - 3 * f(x2), atol=1e-10)


def test_rollout_windows_and_teacher_forcing():
    # a "network" that returns the mean of its input channels + 1 exposes the windowing
    net = lambda z: z.mean(dim=1, keepdim=True) + 1.0
    B, T, ctx, tf = 1, 8, 3, 5
    x = torch.arange(T, dtype=torch.float32).view(1, T, 1, 1, 1).expand(B, T, 1, 2, 2).contiguous()
    out = fno_ref.ns_rollout(net, x, tf, ctx)[0, :, 0, 0, 0]
    exp = []
    for t in range(T):
        if t < ctx - 1:
            exp.append(float(t))                       # context not full: echo the last observation
        else:
            frames = [float(j) if j < tf else exp[j - 1] for j in range(t - ctx + 1, t + 1)]
            exp.append(sum(frames) / ctx + 1.0)
    assert torch.allclose(out, torch.tensor(exp))


def test_single_frame_rollout_equals_context_one():
    net = fno_ref.FNO([4, 4], 1, 4, 8, 8, 1, 2, seed=3)
    x = torch.randn(2, 5, 1, 16, 16)
    a = fno_ref.ns_rollout_single(net, x, 3)
    b = fno_ref.ns_rollout(net, x, 3, 1)
    assert torch.allclose(a, b, atol=1e-6)


def test_mode_major_roundtrip():
    w = torch.randn(3, 5, 4, 3, dtype=torch.complex64)
    assert torch.equal(fno_ref.spec_from_mode_major(fno_ref.spec_to_mode_major(w)), w)


# ---- the in-tree rollout drivers, pinned by vectors produced by EXECUTING the reference's classes
# (tests/golden/make_fno_driver_golden.py: reference TFNO2DModule / FNOModule / FNO2DModule around a stub neuralop FNO)
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fno_driver_golden.npz")


def golden_net(G, tag, in_channels, out_channels):
    """oracle FNO carrying the golden's parameters; returns (net, names)."""
    ctx, _, hidden, layers, m1, m2 = [int(v) for v in G[f"{tag}/cfg"][:6]] if not tag.startswith("dl") else \
        (0, 0, int(G[f"{tag}/cfg"][4]), int(G[f"{tag}/cfg"][5]), int(G[f"{tag}/cfg"][6]), int(G[f"{tag}/cfg"][7]))
    net = fno_ref.FNO([m1, m2], in_channels, hidden, 16, 16, out_channels, layers)
    for name in list(net.params):
        a = torch.from_numpy(G[f"{tag}/p/{name}"])
        net.params[name] = torch.view_as_complex(a.contiguous()) if ".convs.weight." in name else a
    return net


def golden_grad(G, tag, name):
    a = torch.from_numpy(G[f"{tag}/g/{name}"])
    return torch.view_as_complex(a.contiguous()) if ".convs.weight." in name else a


def close(a, b, tol=1e-5):
    if a.is_complex():
        a, b = torch.view_as_real(a), torch.view_as_real(b)
    return ((a.double() - b.double()).abs().max() <= tol * b.double().abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("tag", ["ns_a", "ns_b", "ns_c", "ns_d", "ns_e"])
def test_ns_rollout_restatement_equals_the_reference_driver(tag):
    G = np.load(GOLD)
    ctx, tf = int(G[f"{tag}/cfg"][0]), int(G[f"{tag}/cfg"][1])
    net = golden_net(G, tag, ctx, 1).requires_grad_(True)
    x = torch.from_numpy(G[f"{tag}/x"]).requires_grad_(True)
    y = torch.from_numpy(G[f"{tag}/y"])
    out = fno_ref.ns_rollout(net, x, tf, ctx)
    assert close(out, torch.from_numpy(G[f"{tag}/out"]))
    loss = torch.nn.functional.mse_loss(out, y)
    assert abs(loss.item() - float(G[f"{tag}/loss"])) <= 1e-6 * abs(float(G[f"{tag}/loss"]))
    loss.backward()
    assert close(x.grad, torch.from_numpy(G[f"{tag}/gx"]))
    for name, p in net.params.items():
        assert close(p.grad, golden_grad(G, tag, name)), name


def test_ns_single_frame_restatement_equals_the_reference_driver():
    G = np.load(GOLD)
    net = golden_net(G, "ns_single", 1, 1).requires_grad_(True)
    out = fno_ref.ns_rollout_single(net, torch.from_numpy(G["ns_single/x"]), 3)
    assert close(out, torch.from_numpy(G["ns_single/out"]))
    torch.nn.functional.mse_loss(out, torch.from_numpy(G["ns_single/y"])).backward()
    for name, p in net.params.items():
        assert close(p.grad, golden_grad(G, "ns_single", name)), name


@pytest.mark.parametrize("tag", ["dl_a", "dl_b", "dl_c"])
def test_dlwp_rollout_restatement_equals_the_reference_driver(tag):
    G = np.load(GOLD)
    ctx, Cc, Cp, Cg = [int(v) for v in G[f"{tag}/cfg"][:4]]
    net = golden_net(G, tag, Cc + (Cp + Cg) * ctx, Cg).requires_grad_(True)
    const = torch.from_numpy(G[f"{tag}/constants"]) if Cc else None
    presc = torch.from_numpy(G[f"{tag}/prescribed"]) if Cp else None
    prog = torch.from_numpy(G[f"{tag}/prognostic"])
    out = fno_ref.dlwp_rollout(net, const, presc, prog, ctx)
    assert close(out, torch.from_numpy(G[f"{tag}/out"]))
    # the reference's published forward (one lead time) gave the same first step
    assert close(out[:, :1], torch.from_numpy(G[f"{tag}/one_step"]))
    torch.nn.functional.mse_loss(out, torch.from_numpy(G[f"{tag}/target"])).backward()
    for name, p in net.params.items():
        assert close(p.grad, golden_grad(G, tag, name)), name
