"""GraphedTrainStep (flat parameters + fused gradient accumulation + hipGraph replay) must follow the same
trajectory as the eager per-parameter path: same losses over several optimizer steps, same final parameters."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _afno():
    from dlwp_benchmark_amd import nsbench
    torch.manual_seed(3)
    return nsbench.AFNONet(img_height=16, img_width=16, patch_size=(2, 2), in_chans=1, out_chans=1, embed_dim=32, depth=2,
                           mlp_ratio=2.0, num_blocks=4, context_size=2)


def _swin():
    from dlwp_benchmark_amd import nsbench
    torch.manual_seed(4)
    return nsbench.SwinTransformer(context_size=2, pretrain_img_size=16, patch_size=2, in_chans=1, out_chans=1, embed_dim=16,
                                   depths=[2, 2], num_heads=[2, 2], drop_path_rate=0.0)


@pytest.mark.parametrize("make", [_afno, _swin])
def test_graphed_step_matches_eager(cuda, make):
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep, mse_loss
    g = torch.Generator().manual_seed(11)
    u = torch.randn(2, 7, 1, 16, 16, generator=g).to(cuda)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    call = lambda m, kw: m(kw["x"], 2)   # noqa: E731
    # eager reference trajectory: torch autograd accumulation + torch.optim.Adam on separate parameter tensors
    ref = make().to(cuda).train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    ref_losses = []
    for _ in range(4):
        opt.zero_grad(set_to_none=True)
        loss = mse_loss(ref(x, 2), y)
        loss.backward()
        opt.step()
        ref_losses.append(loss.item())
    for use_graph in (False, True):
        model = make().to(cuda).train()
        step = GraphedTrainStep(model, {"x": x}, y, lr=1e-3, use_graph=use_graph, call=call)
        losses = [step().item() for _ in range(4)]
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 2e-4 * abs(b), (use_graph, losses, ref_losses)
        for (n, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
            assert (p - q).abs().max().item() <= 2e-4, (use_graph, n)


def test_graphed_step_with_allreduce_split_capture(cuda):
    """With a gradient all-reduce the capture is split (forward+backward | all-reduce | optimizer).  One-rank process
    group: the reduce is the identity, so the trajectory must equal the single-graph one."""
    import os
    import torch.distributed as dist
    from dlwp_benchmark_amd import ddp
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        g = torch.Generator().manual_seed(11)
        u = torch.randn(2, 7, 1, 16, 16, generator=g).to(cuda)
        x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
        call = lambda m, kw: m(kw["x"], 2)   # noqa: E731
        plain = GraphedTrainStep(_afno().to(cuda).train(), {"x": x}, y, lr=1e-3, call=call)
        split = GraphedTrainStep(_afno().to(cuda).train(), {"x": x}, y, lr=1e-3, call=call, allreduce=ddp.FlatGradAllReduce())
        for _ in range(3):
            a, b = plain().item(), split().item()
            assert abs(a - b) <= 1e-5 * abs(a), (a, b)
        assert (plain.flat - split.flat).abs().max().item() <= 1e-5
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("use_graph", [False, True])
def test_micro_batch_accumulation_sums_unscaled_gradients(cuda, use_graph):
    """GraphedTrainStep.accumulate / apply (dlwpbench/scripts/train.py:214-233): the gradients of the micro-batches are
    summed un-scaled and one Adam step follows -- equal to eager autograd over the same micro-batches + torch Adam."""
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep, mse_loss
    g = torch.Generator().manual_seed(12)
    u = torch.randn(4, 7, 1, 16, 16, generator=g).to(cuda)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    call = lambda m, kw: m(kw["x"], 2)   # noqa: E731
    ref = _afno().to(cuda).train()
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    opt.zero_grad(set_to_none=True)
    for i in (0, 2):
        mse_loss(ref(x[i:i + 2], 2), y[i:i + 2]).backward()
    ref_grads = {k: p.grad.clone() for k, p in ref.named_parameters() if p.grad is not None}
    opt.step()
    model = _afno().to(cuda).train()
    step = GraphedTrainStep(model, {"x": x[:2]}, y[:2], lr=1e-3, use_graph=use_graph, call=call, graph_optimizer=False)
    for i in (0, 2):
        step.accumulate({"x": x[i:i + 2]}, y[i:i + 2])
    for k, p in model.named_parameters():
        if k not in ref_grads:
            continue
        scale = ref_grads[k].abs().max().clamp_min(1e-12)
        assert ((p.grad - ref_grads[k]).abs().max() / scale).item() <= 2e-4, k
    step.apply()
    for (k, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert ((p - q).abs().max() / q.abs().max().clamp_min(1e-12)).item() <= 2e-4, k
    assert step.grad.abs().max().item() == 0.0          # Adam zeroes the accumulated gradients


def _fno_ns():
    from dlwp_benchmark_amd import nsbench
    torch.manual_seed(7)
    return nsbench.TFNO2DModule(n_modes=[8, 8], in_channels=1, hidden_channels=16, lifting_channels=32, projection_channels=32,
                                out_channels=1, n_layers=2, context_size=2)


def test_fno_module_trains_under_graphed_step(cuda):
    """The FNO rollout modules own ONE flat parameter whose BPTT gradient is produced by the C++ trainer; under
    GraphedTrainStep (flatten_parameters re-points .data / .grad) the trainer must adopt those buffers: same trajectory as
    the module's own fused train_step, and the loss must go down."""
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    g = torch.Generator().manual_seed(11)
    u = torch.randn(2, 7, 1, 32, 32, generator=g).to(cuda)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    ref = _fno_ns().to(cuda)
    opt = ref.make_optimizer(lr=1e-3)
    ref_losses = [ref.train_step(x, y, 3, optimizer=opt).item() for _ in range(5)]
    for use_graph in (False, True):
        model = _fno_ns().to(cuda).train()
        step = GraphedTrainStep(model, {"x": x}, y, lr=1e-3, use_graph=use_graph, call=lambda m, kw: m(kw["x"], 3))
        losses = [step().item() for _ in range(5)]
        assert losses[-1] < losses[0]
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 5e-4 * abs(b), (use_graph, losses, ref_losses)
        assert (model.flat_params.data - ref.flat_params.data).abs().max().item() <= 2e-4


def test_fno_rollout_refuses_backward_after_a_second_forward(cuda):
    """One set of BPTT activations per shape: a second forward before backward must raise instead of silently
    differentiating the wrong rollout (reference nn.Modules keep activations per autograd node)."""
    model = _fno_ns().to(cuda).train()
    x1 = torch.randn(2, 6, 1, 32, 32, device=cuda)
    x2 = torch.randn(2, 6, 1, 32, 32, device=cuda)
    y1 = model(x1, 3)
    with torch.no_grad():
        model(x2, 3)                      # validation-style forward: no activations kept, generation unchanged
    y1.sum().backward()                   # still valid
    y1 = model(x1, 3)
    y2 = model(x2, 3)
    with pytest.raises(RuntimeError, match="another forward"):
        (y1.sum() + y2.sum()).backward()


def test_train_dlwp_with_fno_modules_reduces_loss(cuda):
    """train_loop.train_dlwp (GraphedTrainStep inside) for dlwpbench FNO2DModule and the Tucker TFNO2DModule."""
    import numpy as np
    from dlwp_benchmark_amd import dlwpbench, train_loop, wbdata
    fields, prog, presc, const = wbdata.synthetic_fields(6 * 20 + 8, 16, 32, prognostic={"t2m": [], "z": [500]}, seed=5)
    ds = wbdata.WeatherBenchArrays(fields, prognostic_variable_names_and_levels=prog, prescribed_variable_names=presc,
                                   constant_names=const, sequence_length=4, normalize=True, context_size=1)
    for cls, extra in ((dlwpbench.FNO2DModule, {}), (dlwpbench.TFNO2DModule, {"rank": 0.5})):
        torch.manual_seed(2)
        model = cls(n_modes=[8, 8], constant_channels=4, prescribed_channels=1, prognostic_channels=2, hidden_channels=16,
                    lifting_channels=32, projection_channels=32, n_layers=2, context_size=1, **extra).to(cuda)
        log = train_loop.train_dlwp(model, ds, ds, epochs=4, batch_size=4, learning_rate=2e-3, save_model=False)
        assert np.isfinite(log[-1]["train_mse"])
        assert log[-1]["train_mse"] < log[0]["train_mse"], (cls.__name__, log)


def test_clip_folded_into_adam_matches_clip_then_step(cuda):
    """FusedAdam.step(clip_max_norm=c) (dlwp_adam_step_clipped: the clipping coefficient applied while Adam reads the gradient)
    against clip_grad_norm_() + step() and against torch.nn.utils.clip_grad_norm_ + torch.optim.Adam
    (dlwpbench scripts/train.py:133-136: max_norm = learning rate), with gradients both above and below the threshold."""
    from dlwp_benchmark_amd.fno_engine import FusedAdam
    g = torch.Generator().manual_seed(5)
    n = 100003
    for scale, grad_scale in ((1.0, 1.0), (1e-6, 1.0), (1.0, 0.5)):
        p0 = torch.randn(n, generator=g).to(cuda)
        gr = (torch.randn(n, generator=g) * scale).to(cuda)
        a_p, a_g, b_p, b_g = p0.clone(), gr.clone(), p0.clone(), gr.clone()
        a, b = FusedAdam(a_p, a_g, lr=1e-3), FusedAdam(b_p, b_g, lr=1e-3)
        ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.Adam([ref], lr=1e-3)
        for _ in range(3):
            a_g.copy_(gr)
            b_g.copy_(gr)
            a.step(grad_scale=grad_scale, clip_max_norm=1e-3)
            b.clip_grad_norm_(1e-3, grad_scale=grad_scale)
            b.step(grad_scale=grad_scale)
            ref.grad = gr * grad_scale
            torch.nn.utils.clip_grad_norm_([ref], 1e-3)
            opt.step()
        assert (a_g == 0).all()                                   # zero_grad semantics kept
        assert (a_p - b_p).abs().max().item() <= 1e-6, (scale, grad_scale)
        assert (a_p - ref.data).abs().max().item() <= 2e-6, (scale, grad_scale)
