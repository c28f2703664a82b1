"""GPU parity tests of the FNO rollout path: HIP kernels (through the C ABI) vs the CPU oracle.

Tolerance: BASELINE.json's north_star asks for forward parity <= 1e-4 relative in fp32; the
checks below use a max-norm relative error of 1e-4 for forward values and 5e-4 for gradients
(sums over up to 16k pixels / 11 BPTT steps accumulate fp32 rounding in a different order
than torch's autograd on the CPU).
"""
import ctypes as C

import pytest
import torch

from oracle import fno_ref

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-4
GRAD_TOL = 5e-4


def rel_err(got, ref):
    got, ref = got.detach().cpu(), ref.detach().cpu()
    if got.is_complex() or ref.is_complex():
        got, ref = torch.view_as_real(got.to(torch.complex128)), torch.view_as_real(ref.to(torch.complex128))
    got, ref = got.double(), ref.double()
    return ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.fixture(scope="module")
def L(cuda):
    from dlwp_benchmark_amd import lib
    lib.load()
    return lib


@pytest.mark.parametrize("B,Cin,Ch,Cout,P", [
    (4, 10, 256, 32, 4096),   # lifting at the headline config
    (4, 32, 256, 1, 4096),    # projection at the headline config
    (2, 3, 40, 5, 100),       # ragged: nothing a multiple of 16/64
    (1, 13, 64, 8, 2048),     # dlwp-like channel counts on a 32x64 grid
])
def test_pwmlp_fwd_bwd(L, cuda, B, Cin, Ch, Cout, P):
    lib = L.load()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, Cin, P, generator=g)
    w1 = torch.randn(Ch, Cin, generator=g) / Cin ** 0.5
    b1 = torch.randn(Ch, generator=g) * 0.1
    w2 = torch.randn(Cout, Ch, generator=g) / Ch ** 0.5
    b2 = torch.randn(Cout, generator=g) * 0.1
    gy = torch.randn(B, Cout, P, generator=g)
    ref_in = [t.clone().requires_grad_(True) for t in (x, w1, b1, w2, b2)]
    y_ref = fno_ref.pw_mlp(ref_in[0][..., None], *ref_in[1:])[..., 0]
    y_ref.backward(gy)

    dx, dw1, db1, dw2, db2, dgy = [t.to(cuda) for t in (x, w1, b1, w2, b2, gy)]
    y = torch.empty(B, Cout, P, device=cuda)
    L.check(lib.dlwp_pwmlp_fwd(L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(db2), L.ptr(y),
                               B, Cin, Ch, Cout, P, L.stream()))
    assert rel_err(y, y_ref) <= FWD_TOL

    gx = torch.full_like(dx, float("nan"))
    gw1, gb1, gw2, gb2 = [torch.zeros_like(t) for t in (dw1, db1, dw2, db2)]
    L.check(lib.dlwp_pwmlp_bwd(L.ptr(dx), L.ptr(dw1), L.ptr(db1), L.ptr(dw2), L.ptr(dgy), L.ptr(gx),
                               L.ptr(gw1), L.ptr(gb1), L.ptr(gw2), L.ptr(gb2), B, Cin, Ch, Cout, P, L.stream()))
    torch.cuda.synchronize()
    for got, ref, name in zip((gx, gw1, gb1, gw2, gb2), ref_in, ("gx", "gw1", "gb1", "gw2", "gb2")):
        assert rel_err(got, ref.grad) <= GRAD_TOL, name


@pytest.mark.parametrize("B,Cc,H,W,n_modes,act_in", [
    (4, 32, 64, 64, (12, 12), 0),
    (4, 32, 64, 64, (12, 12), 1),
    (2, 20, 32, 64, (8, 9), 1),      # ragged channels, odd mode counts, non-square grid
    (1, 8, 16, 16, (16, 16), 0),     # all rows kept and the Nyquist column included
    (2, 32, 128, 256, (12, 12), 1),  # C4-sized grid
    # wide layers (hidden_channels > 64: channel-blocked row / spatial kernels, weight slice streamed into MFMA fragments)
    (2, 77, 32, 32, (12, 12), 1),
    (4, 217, 64, 64, (12, 12), 1),
    (3, 100, 32, 64, (8, 9), 0),
    (9, 80, 16, 16, (6, 6), 1),      # more than 8 samples: two MFMA row tiles in the mode contraction
    (12, 32, 32, 32, (8, 8), 1),     # the same on the fused narrow path
])
def test_fno_block_fwd_bwd(L, cuda, B, Cc, H, W, n_modes, act_in):
    lib = L.load()
    m1, m2c = n_modes[0], n_modes[1] // 2 + 1
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, Cc, H, W, generator=g)
    wspec = torch.view_as_complex(torch.randn(Cc, Cc, m1, m2c, 2, generator=g) / Cc ** 0.5)
    wskip = torch.randn(Cc, Cc, generator=g) / Cc ** 0.5
    bias = torch.randn(Cc, generator=g) * 0.1
    gpre = torch.randn(B, Cc, H, W, generator=g)

    xr, wr, kr, br = [t.clone().requires_grad_(True) for t in (x, wspec, wskip, bias)]
    xin = torch.nn.functional.gelu(xr) if act_in else xr
    pre_ref = fno_ref.fno_block(xin, wr, kr, br, list(n_modes))
    pre_ref.backward(gpre)

    plan = C.c_void_p()
    L.check(lib.dlwp_fno_plan_create(Cc, H, W, m1, m2c, C.byref(plan)))
    try:
        ws = torch.empty(lib.dlwp_fno_block_workspace_bytes(plan, B), dtype=torch.uint8, device=cuda)
        dx, dk, db, dg = [t.to(cuda) for t in (x, wskip, bias, gpre)]
        dw = fno_ref.spec_to_mode_major(wspec).to(cuda)
        pre = torch.empty(B, Cc, H, W, device=cuda)
        xhat = torch.empty(B, m1, m2c, Cc, 2, device=cuda)
        L.check(lib.dlwp_fno_block_fwd(plan, L.ptr(dx), act_in, L.ptr(dw), L.ptr(dk), L.ptr(db), L.ptr(pre),
                                       L.ptr(xhat), B, L.ptr(ws), L.stream()))
        assert rel_err(pre, pre_ref) <= FWD_TOL
        gx = torch.full_like(dx, float("nan"))
        gw, gk, gb = torch.zeros_like(dw), torch.zeros_like(dk), torch.zeros_like(db)
        L.check(lib.dlwp_fno_block_bwd(plan, L.ptr(dx), act_in, L.ptr(dw), L.ptr(dk), L.ptr(dg), L.ptr(xhat),
                                       L.ptr(gx), L.ptr(gw), L.ptr(gk), L.ptr(gb), B, L.ptr(ws), L.stream()))
        torch.cuda.synchronize()
        assert rel_err(gx, xr.grad) <= GRAD_TOL, "gx"
        assert rel_err(fno_ref.spec_from_mode_major(gw.cpu()), wr.grad) <= GRAD_TOL, "gwspec"
        assert rel_err(gk, kr.grad) <= GRAD_TOL, "gwskip"
        assert rel_err(gb, br.grad) <= GRAD_TOL, "gbias"
    finally:
        lib.dlwp_fno_plan_destroy(plan)


def _oracle_and_module(cuda, n_modes, D, hidden, lifting, projection, n_layers, ctx, cls_name="TFNO2DModule"):
    from dlwp_benchmark_amd import nsbench
    oracle = fno_ref.FNO(n_modes, D * max(1, ctx), hidden, lifting, projection, D, n_layers, seed=1234)
    kw = dict(n_modes=list(n_modes), in_channels=D, hidden_channels=hidden, lifting_channels=lifting,
              projection_channels=projection, out_channels=D, n_layers=n_layers, type=cls_name, name="t")
    if cls_name == "TFNO2DModule":
        kw["context_size"] = ctx
    module = getattr(nsbench, cls_name)(**kw)
    sd = {}
    for k, v in oracle.params.items():
        if ".convs.weight." in k:
            sd["fno." + k + ".tensor"] = v
        elif ".convs.bias." in k:
            continue
        elif k.endswith("weight"):
            sd["fno." + k] = v[:, :, None, None]
        else:
            sd["fno." + k] = v
    sd["fno.fno_blocks.convs.bias"] = torch.stack(
        [oracle.params[f"fno_blocks.convs.bias.{l}"] for l in range(n_layers)])[:, :, None, None]
    module.load_state_dict(sd)
    return oracle, module.to(cuda)


def _oracle_flat_grad(oracle, module):
    """oracle autograd gradients packed like the module's flat gradient buffer"""
    lay = module.layout
    flat = torch.zeros(lay.total)
    for name in lay.entries:
        gr = oracle.params[name].grad
        dst = lay.view(flat, name)
        if ".convs.weight." in name:
            dst.copy_(fno_ref.spec_to_mode_major(gr))
        else:
            dst.copy_(gr.reshape(dst.shape))
    return flat


@pytest.mark.parametrize("cfg", [
    dict(B=2, T=6, D=1, H=32, W=32, ctx=3, tf=4, hidden=16, lifting=32, projection=32, n_layers=2, n_modes=(8, 8)),
    dict(B=2, T=5, D=2, H=32, W=64, ctx=2, tf=2, hidden=20, lifting=48, projection=24, n_layers=3, n_modes=(6, 10)),
    dict(B=4, T=12, D=1, H=64, W=64, ctx=10, tf=10, hidden=32, lifting=256, projection=256, n_layers=4, n_modes=(12, 12)),
    # edge cases of the rollout windowing (fno.py:217-250): a single net call (T == ctx), fully teacher-forced (tf == T),
    # context 1 with one observed frame, odd batch sizes
    dict(B=1, T=4, D=1, H=32, W=32, ctx=4, tf=4, hidden=16, lifting=32, projection=32, n_layers=1, n_modes=(8, 8)),
    dict(B=3, T=7, D=1, H=32, W=32, ctx=2, tf=7, hidden=16, lifting=32, projection=32, n_layers=2, n_modes=(8, 8)),
    dict(B=5, T=6, D=1, H=32, W=32, ctx=1, tf=1, hidden=8, lifting=16, projection=16, n_layers=2, n_modes=(4, 4)),
    dict(B=2, T=9, D=1, H=16, W=16, ctx=3, tf=3, hidden=24, lifting=40, projection=40, n_layers=2, n_modes=(8, 8)),
])
def test_rollout_train_step_matches_oracle(cuda, cfg):
    oracle, module = _oracle_and_module(cuda, cfg["n_modes"], cfg["D"], cfg["hidden"], cfg["lifting"],
                                        cfg["projection"], cfg["n_layers"], cfg["ctx"])
    g = torch.Generator().manual_seed(99)
    u = torch.randn(cfg["B"], cfg["T"] + 1, cfg["D"], cfg["H"], cfg["W"], generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    oracle.requires_grad_(True)
    loss_ref, yhat_ref = fno_ref.train_step(oracle, x, y, cfg["tf"], cfg["ctx"])

    # (1) inference forward
    with torch.no_grad():
        yhat = module(x.to(cuda), teacher_forcing_steps=cfg["tf"])
    assert rel_err(yhat, yhat_ref) <= FWD_TOL
    # (2) fused eager step: loss + flat gradients
    module.flat_grad.zero_()
    loss = module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=None, use_graph=False)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    gref = _oracle_flat_grad(oracle, module)
    g_eager = module.flat_grad.clone()
    assert rel_err(g_eager, gref) <= GRAD_TOL
    for name in module.layout.entries:  # per-tensor, so a small tensor cannot hide behind a big one
        assert rel_err(module.layout.view(g_eager, name), module.layout.view(gref, name)) <= 2e-3, name
    # (3) hipGraph replay gives the same numbers (twice: capture run and pure replay)
    for _ in range(2):
        module.flat_grad.zero_()
        loss_g = module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=None, use_graph=True)
        torch.cuda.synchronize()
        assert abs(loss_g.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
        assert rel_err(module.flat_grad, gref) <= GRAD_TOL
    # (4) autograd bridge with an arbitrary loss (here: the same MSE through torch)
    module.flat_grad.zero_()
    out = module(x.to(cuda), teacher_forcing_steps=cfg["tf"])
    torch.nn.functional.mse_loss(out, y.to(cuda)).backward()
    assert rel_err(module.flat_params.grad, gref) <= GRAD_TOL


def test_fno_module_single_frame_form(cuda):
    """FNOModule (fno.py:29-41) == context_size-1 rollout; C1 of BASELINE.json."""
    oracle, module = _oracle_and_module(cuda, (12, 12), 1, 32, 256, 256, 4, 1, cls_name="FNOModule")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4, 3, 1, 64, 64, generator=g)
    ref = fno_ref.ns_rollout_single(oracle, x, teacher_forcing_steps=2)
    with torch.no_grad():
        got = module(x.to(cuda), teacher_forcing_steps=2)
    assert rel_err(got, ref) <= FWD_TOL


def test_adam_matches_torch(cuda):
    from dlwp_benchmark_amd.fno_engine import FusedAdam
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(10007, generator=g)
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p = p0.to(cuda)
    grad = torch.zeros_like(p)
    fused = FusedAdam(p, grad, lr=1e-3)
    for it in range(5):
        gi = torch.randn(10007, generator=g)
        p_ref.grad = gi.clone()
        opt.step()
        grad.copy_(gi)
        fused.step()
        assert grad.abs().max().item() == 0.0  # zeroed by the step
    assert rel_err(p, p_ref) <= 1e-6
    # clip_grad_norm_ (train.py:123-125: max_norm = lr)
    gi = torch.randn(10007, generator=g)
    grad.copy_(gi)
    fused.clip_grad_norm_(1e-3)
    ref = gi.clone()
    total = ref.norm()
    ref = ref * min(1.0, 1e-3 / (total.item() + 1e-6))
    assert rel_err(grad, ref) <= 1e-5


def test_training_reduces_loss_and_tracks_oracle(cuda):
    """Three fused steps (graph + FusedAdam) follow the oracle's torch.optim.Adam trajectory."""
    cfg = dict(B=2, T=6, D=1, H=32, W=32, ctx=3, tf=4, hidden=16, lifting=32, projection=32, n_layers=2, n_modes=(8, 8))
    oracle, module = _oracle_and_module(cuda, cfg["n_modes"], cfg["D"], cfg["hidden"], cfg["lifting"],
                                        cfg["projection"], cfg["n_layers"], cfg["ctx"])
    g = torch.Generator().manual_seed(11)
    u = torch.randn(cfg["B"], cfg["T"] + 1, 1, 32, 32, generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    oracle.requires_grad_(True)
    opt_ref = torch.optim.Adam(oracle.parameters(), lr=1e-3)
    opt = module.make_optimizer(lr=1e-3)
    ref_losses, losses = [], []
    for _ in range(3):
        ref_losses.append(fno_ref.train_step(oracle, x, y, cfg["tf"], cfg["ctx"], optimizer=opt_ref)[0].item())
        losses.append(module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=opt).item())
    assert losses[-1] < losses[0]
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-3 * abs(b)


@pytest.mark.parametrize("cfg", [
    dict(B=2, T=5, Cc=4, Cp=1, Cg=5, H=32, W=64, ctx=1, hidden=32, lifting=64, projection=64, n_layers=2, n_modes=(12, 12)),
    dict(B=2, T=6, Cc=2, Cp=1, Cg=3, H=16, W=32, ctx=2, hidden=16, lifting=32, projection=32, n_layers=2, n_modes=(6, 8)),
    dict(B=1, T=4, Cc=0, Cp=0, Cg=2, H=16, W=16, ctx=1, hidden=16, lifting=32, projection=32, n_layers=1, n_modes=(4, 4)),
])
def test_dlwp_form_rollout_matches_oracle(cuda, cfg):
    """dlwpbench FNO2DModule: constants | prescribed | prognostic window, residual output, T-ctx steps."""
    from dlwp_benchmark_amd import dlwpbench
    Cin = cfg["Cc"] + (cfg["Cp"] + cfg["Cg"]) * cfg["ctx"]
    oracle = fno_ref.FNO(cfg["n_modes"], Cin, cfg["hidden"], cfg["lifting"], cfg["projection"], cfg["Cg"],
                         cfg["n_layers"], seed=77)
    module = dlwpbench.FNO2DModule(n_modes=list(cfg["n_modes"]), constant_channels=cfg["Cc"],
                                   prescribed_channels=cfg["Cp"], prognostic_channels=cfg["Cg"],
                                   hidden_channels=cfg["hidden"], lifting_channels=cfg["lifting"],
                                   projection_channels=cfg["projection"], n_layers=cfg["n_layers"],
                                   context_size=cfg["ctx"], type="FNO2DModule", name="t")
    sd = {}
    for k, v in oracle.params.items():
        if ".convs.weight." in k:
            sd["fno." + k + ".tensor"] = v
        elif ".convs.bias." in k:
            continue
        elif k.endswith("weight"):
            sd["fno." + k] = v[:, :, None, None]
        else:
            sd["fno." + k] = v
    sd["fno.fno_blocks.convs.bias"] = torch.stack(
        [oracle.params[f"fno_blocks.convs.bias.{l}"] for l in range(cfg["n_layers"])])[:, :, None, None]
    module.load_state_dict(sd)
    module = module.to(cuda)
    g = torch.Generator().manual_seed(5)
    B, T, H, W = cfg["B"], cfg["T"], cfg["H"], cfg["W"]
    constants = torch.randn(B, 1, cfg["Cc"], H, W, generator=g) if cfg["Cc"] else None
    prescribed = torch.randn(B, T, cfg["Cp"], H, W, generator=g) if cfg["Cp"] else None
    prognostic = torch.randn(B, T, cfg["Cg"], H, W, generator=g)
    target = torch.randn(B, T - cfg["ctx"], cfg["Cg"], H, W, generator=g)
    oracle.requires_grad_(True)
    y_ref = fno_ref.dlwp_rollout(oracle, constants, prescribed, prognostic, cfg["ctx"])
    loss_ref = torch.nn.functional.mse_loss(y_ref, target)
    loss_ref.backward()
    dev = lambda t_: None if t_ is None else t_.to(cuda)
    with torch.no_grad():
        y = module(dev(constants), dev(prescribed), dev(prognostic))
    assert rel_err(y, y_ref) <= FWD_TOL
    gref = _oracle_flat_grad(oracle, module)
    for use_graph in (False, True, True):
        module.flat_grad.zero_()
        loss = module.train_step(dev(constants), dev(prescribed), dev(prognostic), dev(target), use_graph=use_graph)
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
        assert rel_err(module.flat_grad, gref) <= GRAD_TOL


def test_two_live_trainers_do_not_disturb_each_other(cuda):
    """Regression: a hipMemsetAsync node captured into the step graph intermittently wrote garbage instead of zero into
    the loss scalar once a second model had been created in the process (ROCm 7.2); zero fills are kernels now."""
    from dlwp_benchmark_amd import nsbench

    def mk():
        torch.manual_seed(7)
        return nsbench.TFNO2DModule(n_modes=[8, 8], in_channels=1, hidden_channels=16, lifting_channels=32,
                                    projection_channels=32, out_channels=1, n_layers=2, context_size=2).to(cuda)
    g = torch.Generator().manual_seed(0)
    u = torch.randn(4, 9, 1, 32, 32, generator=g).to(cuda)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    for _ in range(3):
        m = mk()
        opt = m.make_optimizer(lr=5e-3)
        ref = [m.train_step(x, y, 4, optimizer=opt).item() for _ in range(3)]
        m2 = mk()
        opt2 = m2.make_optimizer(lr=5e-3)
        m2.flat_params.data.copy_(m.flat_params.data)
        opt2.exp_avg.copy_(opt.exp_avg); opt2.exp_avg_sq.copy_(opt.exp_avg_sq); opt2.step_count.copy_(opt.step_count)   # noqa: E702
        junk = [torch.randn(1 << k, device=cuda) for k in range(4, 16)]      # churn the allocator
        for _ in range(3):
            a = m.train_step(x, y, 4, optimizer=opt).item()
            b = m2.train_step(x, y, 4, optimizer=opt2).item()
            assert abs(a - b) <= 1e-5 * abs(b) and 0 < a < 10, (a, b, ref)
        del junk


# ---- HIP trainer vs vectors produced by executing the reference's own rollout classes
# (tests/golden/make_fno_driver_golden.py; the backbone inside is the oracle's FNO, the drivers are the reference's)
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden", "fno_driver_golden.npz")


def _golden_state_dict(G, tag, n_layers):
    sd, biases = {}, []
    for key in G.files:
        if not key.startswith(f"{tag}/p/"):
            continue
        name = key[len(f"{tag}/p/"):]
        a = torch.from_numpy(G[key])
        if ".convs.weight." in name:
            sd[f"fno.{name}.tensor"] = torch.view_as_complex(a.contiguous())
        elif ".convs.bias." in name:
            continue
        elif name.endswith("weight"):
            sd["fno." + name] = a[:, :, None, None]
        else:
            sd["fno." + name] = a
    sd["fno.fno_blocks.convs.bias"] = torch.stack(
        [torch.from_numpy(G[f"{tag}/p/fno_blocks.convs.bias.{l}"]) for l in range(n_layers)])[:, :, None, None]
    return sd


def _golden_flat_grad(G, tag, module):
    lay = module.layout
    flat = torch.zeros(lay.total)
    for name in lay.entries:
        a = torch.from_numpy(G[f"{tag}/g/{name}"])
        dst = lay.view(flat, name)
        if ".convs.weight." in name:
            dst.copy_(fno_ref.spec_to_mode_major(torch.view_as_complex(a.contiguous())))
        else:
            dst.copy_(a.reshape(dst.shape))
    return flat


@pytest.mark.parametrize("tag", ["ns_a", "ns_b", "ns_c", "ns_d", "ns_e"])
def test_ns_trainer_matches_reference_driver_golden(cuda, tag):
    from dlwp_benchmark_amd import nsbench
    G = np.load(GOLD)
    ctx, tf, hidden, layers, m1, m2 = [int(v) for v in G[f"{tag}/cfg"]]
    module = nsbench.TFNO2DModule(n_modes=[m1, m2], in_channels=1, hidden_channels=hidden, lifting_channels=16,
                                  projection_channels=16, out_channels=1, n_layers=layers, context_size=ctx)
    module.load_state_dict(_golden_state_dict(G, tag, layers))
    module = module.to(cuda)
    x, y = torch.from_numpy(G[f"{tag}/x"]).to(cuda), torch.from_numpy(G[f"{tag}/y"]).to(cuda)
    with torch.no_grad():
        out = module(x, teacher_forcing_steps=tf)
    assert rel_err(out, torch.from_numpy(G[f"{tag}/out"])) <= FWD_TOL
    module.flat_grad.zero_()
    loss = module.train_step(x, y, tf, optimizer=None, use_graph=True)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(G[f"{tag}/loss"])) <= 1e-4 * abs(float(G[f"{tag}/loss"]))
    assert rel_err(module.flat_grad, _golden_flat_grad(G, tag, module)) <= GRAD_TOL


def test_ns_single_frame_trainer_matches_reference_driver_golden(cuda):
    from dlwp_benchmark_amd import nsbench
    G = np.load(GOLD)
    module = nsbench.FNOModule(n_modes=[6, 6], in_channels=1, hidden_channels=8, lifting_channels=16, projection_channels=16,
                               out_channels=1, n_layers=2)
    module.load_state_dict(_golden_state_dict(G, "ns_single", 2))
    module = module.to(cuda)
    x, y = torch.from_numpy(G["ns_single/x"]).to(cuda), torch.from_numpy(G["ns_single/y"]).to(cuda)
    with torch.no_grad():
        assert rel_err(module(x, teacher_forcing_steps=3), torch.from_numpy(G["ns_single/out"])) <= FWD_TOL
    module.flat_grad.zero_()
    module.train_step(x, y, 3, optimizer=None, use_graph=False)
    assert rel_err(module.flat_grad, _golden_flat_grad(G, "ns_single", module)) <= GRAD_TOL


@pytest.mark.parametrize("tag", ["dl_a", "dl_b", "dl_c"])
def test_dlwp_trainer_matches_reference_driver_golden(cuda, tag):
    from dlwp_benchmark_amd import dlwpbench
    G = np.load(GOLD)
    ctx, Cc, Cp, Cg, hidden, layers, m1, m2 = [int(v) for v in G[f"{tag}/cfg"]]
    module = dlwpbench.FNO2DModule(n_modes=[m1, m2], constant_channels=Cc, prescribed_channels=Cp, prognostic_channels=Cg,
                                   hidden_channels=hidden, lifting_channels=16, projection_channels=16, n_layers=layers,
                                   context_size=ctx)
    module.load_state_dict(_golden_state_dict(G, tag, layers))
    module = module.to(cuda)
    dev = lambda k: torch.from_numpy(G[f"{tag}/{k}"]).to(cuda) if f"{tag}/{k}" in G.files else None   # noqa: E731
    const, presc, prog, target = dev("constants"), dev("prescribed"), dev("prognostic"), dev("target")
    with torch.no_grad():
        out = module(const, presc, prog)
    assert rel_err(out, torch.from_numpy(G[f"{tag}/out"])) <= FWD_TOL
    module.flat_grad.zero_()
    loss = module.train_step(const, presc, prog, target, use_graph=False)
    torch.cuda.synchronize()
    assert abs(loss.item() - float(G[f"{tag}/loss"])) <= 1e-4 * abs(float(G[f"{tag}/loss"]))
    assert rel_err(module.flat_grad, _golden_flat_grad(G, tag, module)) <= GRAD_TOL


def test_headline_config_parity_at_the_benchmarked_horizon(cuda):
    """BASELINE configs[1] exactly as bench.py runs it: TFNO2DModule 64x64, B=4, T=20, context 10, teacher forcing 10 ->
    11 net calls, 10 of them closed loop.  Forward <= 1e-4 relative (north_star) after ten closed-loop steps of error
    growth; loss and gradients against the oracle's autograd."""
    cfg = dict(B=4, T=20, D=1, H=64, W=64, ctx=10, tf=10, hidden=32, lifting=256, projection=256, n_layers=4, n_modes=(12, 12))
    oracle, module = _oracle_and_module(cuda, cfg["n_modes"], cfg["D"], cfg["hidden"], cfg["lifting"], cfg["projection"],
                                        cfg["n_layers"], cfg["ctx"])
    g = torch.Generator().manual_seed(1234)
    u = torch.randn(cfg["B"], cfg["T"] + 1, 1, 64, 64, generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    oracle.requires_grad_(True)
    loss_ref, yhat_ref = fno_ref.train_step(oracle, x, y, cfg["tf"], cfg["ctx"])
    with torch.no_grad():
        yhat = module(x.to(cuda), teacher_forcing_steps=cfg["tf"])
    assert rel_err(yhat, yhat_ref) <= FWD_TOL
    assert rel_err(yhat[:, -1], yhat_ref[:, -1]) <= FWD_TOL          # the last (10th closed-loop) frame on its own
    module.flat_grad.zero_()
    loss = module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=None, use_graph=True)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    assert rel_err(module.flat_grad, _oracle_flat_grad(oracle, module)) <= GRAD_TOL


# the hidden widths of the reference's published TFNO2D sweep (src/nsbench/scripts/train_commands.txt:83-91: 2, 8, 27, 38,
# 54, 77, 108, 154, 217 <-> 5k .. 32M parameters, plot_results.py:58); 4 layers, 12 x 12 modes, lifting / projection 256
@pytest.mark.parametrize("hidden", [2, 8, 27, 38, 54, 77, 108, 154, 217])
def test_published_sweep_widths_match_oracle(cuda, hidden):
    cfg = dict(B=2, T=4, D=1, H=32, W=32, ctx=2, tf=2, hidden=hidden, lifting=256, projection=256, n_layers=4, n_modes=(12, 12))
    oracle, module = _oracle_and_module(cuda, cfg["n_modes"], cfg["D"], cfg["hidden"], cfg["lifting"], cfg["projection"],
                                        cfg["n_layers"], cfg["ctx"])
    g = torch.Generator().manual_seed(1000 + hidden)
    u = torch.randn(cfg["B"], cfg["T"] + 1, 1, cfg["H"], cfg["W"], generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    oracle.requires_grad_(True)
    loss_ref, yhat_ref = fno_ref.train_step(oracle, x, y, cfg["tf"], cfg["ctx"])
    with torch.no_grad():
        yhat = module(x.to(cuda), teacher_forcing_steps=cfg["tf"])
    assert rel_err(yhat, yhat_ref) <= FWD_TOL
    gref = _oracle_flat_grad(oracle, module)
    for use_graph in (False, True):
        module.flat_grad.zero_()
        loss = module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=None, use_graph=use_graph)
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
        assert rel_err(module.flat_grad, gref) <= GRAD_TOL
        for name in module.layout.entries:
            assert rel_err(module.layout.view(module.flat_grad, name), module.layout.view(gref, name)) <= 2e-3, name


def test_wide_dlwp_form_matches_oracle(cuda):
    """dlwpbench FNO2DModule at hidden 96 (wide path): constants / prescribed / prognostic gather, residual output, closed loop."""
    from dlwp_benchmark_amd import dlwpbench
    cfg = dict(B=2, T=4, Cc=4, Cp=1, Cg=5, H=32, W=64, ctx=1, hidden=96, lifting=128, projection=128, n_layers=2, n_modes=(12, 12))
    Cin = cfg["Cc"] + (cfg["Cp"] + cfg["Cg"]) * cfg["ctx"]
    oracle = fno_ref.FNO(cfg["n_modes"], Cin, cfg["hidden"], cfg["lifting"], cfg["projection"], cfg["Cg"], cfg["n_layers"], seed=78)
    module = dlwpbench.FNO2DModule(n_modes=list(cfg["n_modes"]), constant_channels=cfg["Cc"], prescribed_channels=cfg["Cp"],
                                   prognostic_channels=cfg["Cg"], hidden_channels=cfg["hidden"], lifting_channels=cfg["lifting"],
                                   projection_channels=cfg["projection"], n_layers=cfg["n_layers"], context_size=cfg["ctx"])
    sd = {}
    for k, v in oracle.params.items():
        if ".convs.weight." in k:
            sd["fno." + k + ".tensor"] = v
        elif ".convs.bias." in k:
            continue
        elif k.endswith("weight"):
            sd["fno." + k] = v[:, :, None, None]
        else:
            sd["fno." + k] = v
    sd["fno.fno_blocks.convs.bias"] = torch.stack(
        [oracle.params[f"fno_blocks.convs.bias.{l}"] for l in range(cfg["n_layers"])])[:, :, None, None]
    module.load_state_dict(sd)
    module = module.to(cuda)
    g = torch.Generator().manual_seed(6)
    B, T, H, W = cfg["B"], cfg["T"], cfg["H"], cfg["W"]
    constants = torch.randn(B, 1, cfg["Cc"], H, W, generator=g)
    prescribed = torch.randn(B, T, cfg["Cp"], H, W, generator=g)
    prognostic = torch.randn(B, T, cfg["Cg"], H, W, generator=g)
    target = torch.randn(B, T - cfg["ctx"], cfg["Cg"], H, W, generator=g)
    oracle.requires_grad_(True)
    y_ref = fno_ref.dlwp_rollout(oracle, constants, prescribed, prognostic, cfg["ctx"])
    loss_ref = torch.nn.functional.mse_loss(y_ref, target)
    loss_ref.backward()
    with torch.no_grad():
        y = module(constants.to(cuda), prescribed.to(cuda), prognostic.to(cuda))
    assert rel_err(y, y_ref) <= FWD_TOL
    gref = _oracle_flat_grad(oracle, module)
    module.flat_grad.zero_()
    loss = module.train_step(constants.to(cuda), prescribed.to(cuda), prognostic.to(cuda), target.to(cuda), use_graph=True)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    assert rel_err(module.flat_grad, gref) <= GRAD_TOL


@pytest.mark.parametrize("cfg", [
    dict(B=2, T=6, D=1, H=16, W=16, n_modes=(4, 6, 6), hidden=8, lifting=16, projection=16, n_layers=2, tf=5),
    dict(B=1, T=5, D=2, H=16, W=32, n_modes=(2, 4, 8), hidden=12, lifting=16, projection=16, n_layers=1, tf=3),
    dict(B=2, T=8, D=1, H=16, W=16, n_modes=(6, 16, 16), hidden=72, lifting=32, projection=32, n_layers=2, tf=6),
])
def test_fno_context_module_3d_matches_oracle(cuda, cfg):
    """nsbench FNOContextModule (3-D FNO over the context volume, last time slice out; fno.py:44-100) against the oracle's
    rfftn-based restatement: rollout outputs, loss and every gradient."""
    from dlwp_benchmark_amd import nsbench
    nm, D = list(cfg["n_modes"]), cfg["D"]
    ctx = nm[0]
    oracle = fno_ref.FNO3d(nm, D, cfg["hidden"], cfg["lifting"], cfg["projection"], D, cfg["n_layers"], seed=321)
    module = nsbench.FNOContextModule(n_modes=nm, in_channels=D, hidden_channels=cfg["hidden"], lifting_channels=cfg["lifting"],
                                      projection_channels=cfg["projection"], out_channels=D, n_layers=cfg["n_layers"],
                                      context_size=10, type="FNOContextModule", name="t")
    assert module.context_size == ctx
    sd = {}
    for k, v in oracle.params.items():
        if ".convs.weight." in k:
            sd["fno." + k + ".tensor"] = v
        elif ".convs.bias." in k:
            continue
        elif k.endswith("weight"):
            sd["fno." + k] = v[:, :, None, None, None]
        else:
            sd["fno." + k] = v
    sd["fno.fno_blocks.convs.bias"] = torch.stack(
        [oracle.params[f"fno_blocks.convs.bias.{l}"] for l in range(cfg["n_layers"])])[:, :, None, None, None]
    module.load_state_dict(sd)
    back = module.state_dict()
    assert back["fno.fno_blocks.convs.weight.0.tensor"].shape == oracle.params["fno_blocks.convs.weight.0"].shape
    module = module.to(cuda)
    g = torch.Generator().manual_seed(17)
    u = torch.randn(cfg["B"], cfg["T"] + 1, D, cfg["H"], cfg["W"], generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    oracle.requires_grad_(True)
    y_ref = fno_ref.ns_rollout_context(oracle, x, cfg["tf"], ctx)
    loss_ref = torch.nn.functional.mse_loss(y_ref, y)
    loss_ref.backward()
    with torch.no_grad():
        yh = module(x.to(cuda), teacher_forcing_steps=cfg["tf"])
    assert rel_err(yh, y_ref) <= FWD_TOL
    lay = module.layout
    gref = torch.zeros(lay.total)
    for name in lay.entries:
        gr = oracle.params[name].grad
        dst = lay.view(gref, name)
        if ".convs.weight." in name:
            Ci, Co = gr.shape[:2]
            dst.copy_(fno_ref.spec_to_mode_major(gr.reshape(Ci, Co, -1, gr.shape[-1])))
        else:
            dst.copy_(gr.reshape(dst.shape))
    for use_graph in (False, True):
        module.flat_grad.zero_()
        loss = module.train_step(x.to(cuda), y.to(cuda), cfg["tf"], optimizer=None, use_graph=use_graph)
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
        assert rel_err(module.flat_grad, gref) <= GRAD_TOL
        for name in lay.entries:
            assert rel_err(lay.view(module.flat_grad, name), lay.view(gref, name)) <= 2e-3, name
