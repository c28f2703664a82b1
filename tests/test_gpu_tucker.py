"""Tucker (TFNO) weights: HIP mode-product kernels vs the einsum oracle (forward + gradients), and the
TFNO2DModule end to end against the dense oracle run on the reconstructed weights."""
import pytest
import torch

from oracle import fno_ref, tucker_ref

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def test_rank_rule_matches_oracle():
    from dlwp_benchmark_amd.tucker import tucker_rank
    for shape in [(32, 32, 12, 7), (16, 16, 8, 5), (8, 24, 6, 4)]:
        for rank in (0.8, 0.5, 0.05, 1.0):
            assert tucker_rank(shape, rank) == tucker_ref.tucker_rank(shape, rank)


def test_tucker_dense_and_gradients(cuda):
    from dlwp_benchmark_amd.tucker import TuckerSpectralWeight
    torch.manual_seed(3)
    tw = TuckerSpectralWeight(12, 10, 6, 4, 0.6, 0.2).to(cuda)
    core = torch.view_as_complex(tw.core.detach().cpu().contiguous()).clone().requires_grad_(True)
    facs = [torch.view_as_complex(f.detach().cpu().contiguous()).clone().requires_grad_(True) for f in tw.factors]
    ref = tucker_ref.tucker_dense(core, facs)
    got = tw.dense()
    assert rel(got, torch.view_as_real(ref)) <= 1e-5
    g = torch.randn(*ref.shape, 2)
    torch.view_as_real(ref).backward(g)
    got.backward(g.to(cuda))
    assert rel(tw.core.grad, torch.view_as_real(core.grad)) <= 1e-4
    for f, fr in zip(tw.factors, facs):
        assert rel(f.grad, torch.view_as_real(fr.grad)) <= 1e-4


def test_tfno_module_train_step_matches_dense_oracle(cuda):
    from dlwp_benchmark_amd import dlwpbench
    cfg = dict(n_modes=[6, 8], constant_channels=2, prescribed_channels=1, prognostic_channels=3, hidden_channels=16,
               lifting_channels=32, projection_channels=32, n_layers=2, rank=0.6, context_size=1)
    torch.manual_seed(11)
    m = dlwpbench.TFNO2DModule(**cfg, type="TFNO2DModule", name="t").to(cuda)
    # oracle: dense FNO carrying the module's non-spectral parameters and the reconstructed spectral weights,
    # with the Tucker factors as the leaves (gradients flow through tucker_dense)
    Cin = 2 + (1 + 3) * 1
    oracle = fno_ref.FNO(cfg["n_modes"], Cin, 16, 32, 32, 3, 2, seed=1)
    sd = m.state_dict()
    cores, facs = [], []
    for l, tw in enumerate(m.tucker):
        cores.append(torch.view_as_complex(tw.core.detach().cpu().contiguous()).clone().requires_grad_(True))
        facs.append([torch.view_as_complex(f.detach().cpu().contiguous()).clone().requires_grad_(True) for f in tw.factors])
    for k in list(oracle.params):
        if ".convs.weight." in k:
            l = int(k.rsplit(".", 1)[1])
            oracle.params[k] = tucker_ref.tucker_dense(cores[l], facs[l])
        elif ".convs.bias." in k:
            l = int(k.rsplit(".", 1)[1])
            oracle.params[k] = sd["fno.fno_blocks.convs.bias"][l].reshape(-1).cpu().clone().requires_grad_(True)
        else:
            v = sd["fno." + k].cpu()
            oracle.params[k] = (v.reshape(v.shape[0], -1) if v.dim() == 4 else v).clone().requires_grad_(True)
    g = torch.Generator().manual_seed(2)
    B, T, H, W = 2, 4, 16, 32
    constants = torch.randn(B, 1, 2, H, W, generator=g)
    prescribed = torch.randn(B, T, 1, H, W, generator=g)
    prognostic = torch.randn(B, T, 3, H, W, generator=g)
    target = torch.randn(B, T - 1, 3, H, W, generator=g)
    y_ref = fno_ref.dlwp_rollout(oracle, constants, prescribed, prognostic, 1)
    loss_ref = torch.nn.functional.mse_loss(y_ref, target)
    loss_ref.backward()
    dev = lambda t_: t_.to(cuda)
    loss = m.train_step(dev(constants), dev(prescribed), dev(prognostic), dev(target), optimizer=None)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_ref.item()) <= 1e-4 * abs(loss_ref.item())
    for l, tw in enumerate(m.tucker):
        assert rel(tw.core.grad, torch.view_as_real(cores[l].grad)) <= 2e-3, l
        for f, fr in zip(tw.factors, facs[l]):
            assert rel(f.grad, torch.view_as_real(fr.grad)) <= 2e-3, l
    # a non-spectral parameter for good measure
    assert rel(m.layout.view(m.flat_grad, "lifting.fcs.0.weight"), oracle.params["lifting.fcs.0.weight"].grad) <= 5e-4


def test_tfno_clip_is_one_global_norm_over_flat_parameters_and_tucker_factors(cuda):
    """clip_gradients: True is the dlwpbench default (configs/training/default.yaml:3; train.py:230-232 clips at max_norm =
    learning rate over model.parameters()): the norm runs over the non-derived slices of the flat gradient AND the Tucker
    cores / factors, and every buffer is scaled by the same coefficient."""
    from dlwp_benchmark_amd import dlwpbench
    cfg = dict(n_modes=[6, 8], constant_channels=2, prescribed_channels=1, prognostic_channels=3, hidden_channels=16,
               lifting_channels=32, projection_channels=32, n_layers=2, rank=0.6, context_size=1)
    g = torch.Generator().manual_seed(2)
    B, T, H, W = 2, 4, 16, 32
    batch = [torch.randn(B, 1, 2, H, W, generator=g).to(cuda), torch.randn(B, T, 1, H, W, generator=g).to(cuda),
             torch.randn(B, T, 3, H, W, generator=g).to(cuda), torch.randn(B, T - 1, 3, H, W, generator=g).to(cuda)]

    def grads_after(clip):
        torch.manual_seed(11)
        m = dlwpbench.TFNO2DModule(**cfg).to(cuda)
        opt = m.make_optimizer(lr=0.0)                 # lr 0: Adam leaves the parameters alone, zero_grad is the only side effect
        m.train_step(*batch, optimizer=None)
        if clip is not None:
            opt.clip_grad_norm_(clip)
        derived = torch.zeros_like(m.flat_grad, dtype=torch.bool)
        for name in m._spec_names():
            m.layout.view(derived, name).fill_(True)
        assert float(m.flat_grad[derived].abs().max()) == 0.0          # handed to the factors, then zeroed
        return [m.flat_grad.clone()] + [p.grad.clone() for p in m.tucker.parameters()]

    raw = grads_after(None)
    total = torch.sqrt(sum((t.double() ** 2).sum() for t in raw)).item()
    max_norm = 0.25 * total
    got = grads_after(max_norm)
    coef = max_norm / (total + 1e-6)
    for a, b in zip(got, raw):
        assert rel(a, b * coef) <= 1e-5
    # a threshold above the norm leaves everything alone
    for a, b in zip(grads_after(2.0 * total), raw):
        assert rel(a, b) <= 1e-6
    # and the public train_step route accepts the threshold (it raised NotImplementedError before)
    torch.manual_seed(11)
    m = dlwpbench.TFNO2DModule(**cfg).to(cuda)
    m.train_step(*batch, optimizer=m.make_optimizer(lr=1e-3), clip_max_norm=1e-3)
    torch.cuda.synchronize()
