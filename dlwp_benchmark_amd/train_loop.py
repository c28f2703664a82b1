"""The caller of the hot path: the nsbench training script's epoch loop, validation and checkpoint policy
(src/nsbench/scripts/train.py:57-175, utils/utils.py:11-39) around the fused train step.

What is kept: Adam(lr) with CosineAnnealingLR(T_max = epochs) stepped once per epoch (:72-73,171), MSE criterion, the
validation pass without gradients (:137-147), the `_last` / `_best` checkpoint policy (:150-156: a new best validation
error writes `<name>_best.ckpt`, anything else and the final epoch write `<name>_last.ckpt`), the checkpoint dictionary
(model_state_dict, optimizer_state_dict, scheduler_state_dict, epoch + 1, iteration, best_val_error) and resuming from
`_last` (:78-87).  What differs: batches come from the seeded rank-sharded permutation of ddp.py instead of a shuffling
DataLoader, the step is the captured hipGraph, scalars are returned / printed instead of written to TensorBoard, and the
checkpoint is written synchronously (the reference's writer thread reads live parameters while the next epoch trains,
App. B-10).
"""
import json
import math
import os
import time
import warnings

import numpy as np
import torch

from . import ddp
from .evaluate import error_moments


class ScalarLog:
    """The scalars the reference sends to TensorBoard -- tags "Epoch", "Learning Rate", "MSE/training" (per iteration) and
    "MSE/validation" (per epoch), each with global_step = the iteration counter (nsbench/scripts/train.py:104-106,128,147;
    dlwpbench/scripts/train.py:107-108,140,163) -- as JSON lines `<out_dir>/<name>/tensorboard/scalars.jsonl`, one
    {"tag", "value", "step", "wall_time"} object per add_scalar call (tensorboard is not installed in this image; the file
    converts one to one).  Device scalars are kept as tensors and written at `flush()` (end of an epoch): no host
    synchronisation inside the training loop."""

    def __init__(self, out_dir, name, enabled=True):
        self.path = os.path.join(out_dir, name, "tensorboard", "scalars.jsonl") if enabled else None
        self._rows = []
        if self.path:
            os.makedirs(os.path.dirname(self.path), exist_ok=True)

    def add_scalar(self, tag, scalar_value, global_step):
        if self.path:
            self._rows.append((tag, scalar_value, int(global_step), time.time()))

    def flush(self):
        if not self.path or not self._rows:
            return
        with open(self.path, "a") as f:
            for tag, v, step, wall in self._rows:
                f.write(json.dumps({"tag": tag, "value": float(v.item() if torch.is_tensor(v) else v), "step": step,
                                    "wall_time": wall}) + "\n")
        self._rows = []


def cosine_lr(base_lr, epoch, t_max, eta_min=0.0):
    """torch.optim.lr_scheduler.CosineAnnealingLR in closed form."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * epoch / t_max)) / 2


def _named_moment_views(model, opt):
    """[(name, exp_avg view, exp_avg_sq view)] per named parameter, in the reference's shapes: the flat Adam moments of
    fno_engine.FusedAdam sliced the way the parameters are laid out in the flat buffer."""
    if hasattr(model, "layout") and hasattr(model, "flat_params") and opt.exp_avg.numel() == model.flat_params.numel():
        a = model.layout.to_state_dict(opt.exp_avg, prefix="fno.")
        b = model.layout.to_state_dict(opt.exp_avg_sq, prefix="fno.")
        return [(k, a[k], b[k]) for k in a]
    base, out = opt.params.data_ptr(), []
    for name, p in model.named_parameters():
        off = (p.data.data_ptr() - base) // 4
        if not (0 <= off and off + p.numel() <= opt.params.numel()):
            continue
        out.append((name, opt.exp_avg[off:off + p.numel()].view(p.shape), opt.exp_avg_sq[off:off + p.numel()].view(p.shape)))
    return out


def _optimizer_state(model, opt):
    """torch.optim.Adam.state_dict() layout (state / param_groups per parameter index, utils.py:33-39 stores exactly
    optimizer.state_dict()), built from the flat moments; `param_names` (extra key, ignored by torch) records which
    parameter each index is."""
    opt = getattr(opt, "main", opt)          # dlwpbench TFNO2DModule: flat part of the composite optimizer
    views = _named_moment_views(model, opt)
    step = float(opt.step_count.item())
    state = {i: {"step": torch.tensor(step), "exp_avg": a.detach().cpu().clone(), "exp_avg_sq": b.detach().cpu().clone()}
             for i, (_, a, b) in enumerate(views)}
    group = {"lr": opt.lr, "betas": tuple(opt.betas), "eps": opt.eps, "weight_decay": 0, "amsgrad": False, "maximize": False,
             "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(len(views)))}
    return {"state": state, "param_groups": [group], "param_names": [n for n, _, _ in views]}


def write_checkpoint(model, optimizer, scheduler_state, epoch, iteration, best_val_error, dst_path):
    """utils.write_checkpoint (:11-39): same keys; the optimizer entry has torch.optim.Adam's state_dict layout."""
    os.makedirs(os.path.dirname(dst_path), exist_ok=True)
    torch.save({"model_state_dict": {k: v.cpu() for k, v in model.state_dict().items()},
                "optimizer_state_dict": _optimizer_state(model, optimizer), "scheduler_state_dict": scheduler_state,
                "epoch": epoch + 1, "iteration": iteration, "best_val_error": best_val_error}, dst_path)


def _load_optimizer_state(st, model, optimizer, path):
    optimizer = getattr(optimizer, "main", optimizer)
    if "exp_avg" in st:                                   # round-1 flat layout
        optimizer.exp_avg.copy_(st["exp_avg"])
        optimizer.exp_avg_sq.copy_(st["exp_avg_sq"])
        optimizer.step_count.copy_(st["step"])
        return True
    if not st.get("state"):
        return False
    views = _named_moment_views(model, optimizer)
    names = st.get("param_names")
    # reference files carry no names: parameter index = registration order, which the drop-in modules keep
    index = {n: i for i, n in enumerate(names)} if names else {n: i for i, (n, _, _) in enumerate(views)}
    found = {}
    for name, a, _ in views:
        ent = st["state"].get(index.get(name, -1))
        if ent is not None and tuple(ent["exp_avg"].shape) == tuple(a.shape):
            found[name] = ent
    if len(found) != len(views):
        warnings.warn(f"{path}: optimizer state matches {len(found)} of {len(views)} parameters; the others restart from "
                      "zero moments")
    if not found:
        return False
    if hasattr(model, "layout") and hasattr(model, "flat_params"):
        if len(found) != len(views):
            return False                                  # the FNO converter needs every tensor
        # the FNO layout converts (complex, mode-major) on the way out: write back through the same converter
        model.layout.from_state_dict(optimizer.exp_avg, {n: e["exp_avg"].to(optimizer.exp_avg.device) for n, e in found.items()},
                                     prefix="fno.")
        model.layout.from_state_dict(optimizer.exp_avg_sq,
                                     {n: e["exp_avg_sq"].to(optimizer.exp_avg.device) for n, e in found.items()}, prefix="fno.")
    else:
        for name, a, b in views:
            if name in found:
                a.copy_(found[name]["exp_avg"].to(a.device))
                b.copy_(found[name]["exp_avg_sq"].to(b.device))
    optimizer.step_count.fill_(int(max(float(e["step"]) for e in found.values())))
    return True


def load_checkpoint(path, model, optimizer=None):
    """Reads this package's checkpoints and the reference's (utils.py:33-39: tensors, scalars and plain containers only, so
    the restricted unpickler is enough)."""
    ck = torch.load(path, map_location="cpu", weights_only=True)
    model.load_state_dict(ck["model_state_dict"])
    if optimizer is not None:
        st = ck.get("optimizer_state_dict")
        if not (isinstance(st, dict) and _load_optimizer_state(st, model, optimizer, path)):
            warnings.warn(f"{path}: no usable optimizer state; Adam restarts from zero moments and step 0")
    return ck


@torch.no_grad()
def validation_mse(model, u_val, sequence_length, batch_size, teacher_forcing_steps, device):
    """MSE over every validation sample's first window (train.py:137-147 concatenates all outputs, then one MSE)."""
    tot, cnt = 0.0, 0
    for i0 in range(0, u_val.shape[0], batch_size):
        seq = u_val[i0:i0 + batch_size, :sequence_length].to(device)
        x, y = seq[:, :-1].contiguous(), seq[:, 1:].contiguous()
        y_hat = model(x, teacher_forcing_steps)
        B, T, D, H, W = y_hat.shape
        m = error_moments(y_hat.reshape(B, T, D * H, W), y.reshape(B, T, D * H, W))
        tot += m[0].sum().item()
        cnt += y_hat.numel()
    return tot / max(cnt, 1)


def train_ns(model, u_train, u_val, name="model", epochs=10, batch_size=4, sequence_length=21, learning_rate=1e-3,
             teacher_forcing_steps=10, val_teacher_forcing_steps=None, noise=0.0, clip_gradients=False, seed=1234,
             out_dir="outputs", save_model=True, continue_training=False, verbose=False, log_scalars=True, stop_epoch=None):
    """Train an nsbench FNO-family module (anything with make_optimizer / train_step) on trajectories u [N, T, D, H, W].
    Returns a list of per-epoch dicts(epoch, lr, train_mse, val_mse).  stop_epoch: leave after that many epochs of the
    `epochs`-long schedule with a `_last` checkpoint written (a long run split over several processes: continue_training)."""
    device = next(model.parameters()).device
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    opt = model.make_optimizer(lr=learning_rate)
    reducer = ddp.FlatGradAllReduce() if world > 1 else None
    ckpt_last = os.path.join(out_dir, name, "checkpoints", f"{name}_last.ckpt")
    epoch0, iteration, best = 0, 0, float("inf")
    if continue_training:
        ck = load_checkpoint(ckpt_last, model, opt)
        epoch0, iteration, best = ck["epoch"], ck["iteration"], ck["best_val_error"]
    if world > 1:
        # replicas must start identical: the module initialises from the process-global RNG, and a resumed rank 0 must
        # hand its Adam state to the others as well (the all-reduced gradient is applied by every rank)
        ddp.broadcast_parameters(model.flat_params.data, src=0)
        o = getattr(opt, "main", opt)
        for buf in (o.exp_avg, o.exp_avg_sq, o.step_count):
            ddp.broadcast_parameters(buf, src=0)
    vtf = teacher_forcing_steps if val_teacher_forcing_steps is None else val_teacher_forcing_steps
    log = []
    writer = ScalarLog(out_dir, name, enabled=log_scalars and save_model and rank == 0)
    for epoch in range(epoch0, epochs):
        opt.lr = cosine_lr(learning_rate, epoch, epochs)
        writer.add_scalar("Epoch", epoch, iteration)
        writer.add_scalar("Learning Rate", opt.lr, iteration)
        losses = []
        for idx in ddp.shard_indices(u_train.shape[0], epoch, rank, world, batch_size, seed):
            xs, ys = zip(*(ddp.ns_sample(u_train, int(i), epoch, sequence_length, noise, seed) for i in idx))
            x, y = torch.stack(xs).to(device), torch.stack(ys).to(device)
            loss = model.train_step(x, y, teacher_forcing_steps, optimizer=opt, grad_scale=1.0 / world, allreduce=reducer,
                                    clip_max_norm=opt.lr if clip_gradients else None)     # clip threshold = lr (:123-125)
            losses.append(loss.clone())   # the step returns its persistent loss buffer; clones stay on the device (no sync)
            writer.add_scalar("MSE/training", losses[-1], iteration)
            iteration += 1
        train_mse = torch.stack(losses).mean().item() if losses else float("nan")
        val_mse = validation_mse(model, u_val, sequence_length, batch_size, vtf, device)
        writer.add_scalar("MSE/validation", val_mse, iteration)
        writer.flush()
        if save_model and rank == 0:
            sched = {"T_max": epochs, "last_epoch": epoch, "base_lrs": [learning_rate], "_last_lr": [opt.lr]}
            if val_mse > best or epoch == epochs - 1:
                dst = ckpt_last
            else:
                best, dst = val_mse, ckpt_last.replace("last", "best")
            write_checkpoint(model, opt, sched, epoch, iteration, best, dst)
            if stop_epoch is not None and epoch + 1 >= stop_epoch and dst != ckpt_last:
                write_checkpoint(model, opt, sched, epoch, iteration, best, ckpt_last)     # the resume point must be current
        log.append({"epoch": epoch, "lr": opt.lr, "train_mse": train_mse, "val_mse": val_mse})
        if verbose and rank == 0:
            print(f"Epoch {str(epoch).zfill(3)}/{epochs}\tMSE train: {train_mse:.2E}\tMSE val: {val_mse:.2E}")
        if stop_epoch is not None and epoch + 1 >= stop_epoch:
            break
    return log


@torch.no_grad()
def validation_mse_dlwp(model, dataset, batch_size, device):
    """dlwpbench/scripts/train.py:236-252: MSE over the concatenated outputs of every validation sample."""
    from . import wbdata
    tot, cnt = 0.0, 0
    was_training = model.training
    model.eval()
    for i0 in range(0, len(dataset), batch_size):
        c, p, g, t = wbdata.to_device_batch([dataset[i] for i in range(i0, min(i0 + batch_size, len(dataset)))], device)
        y_hat = model(constants=c, prescribed=p, prognostic=g)
        B, T, D, H, W = y_hat.shape
        m = error_moments(y_hat.reshape(B, T, D * H, W), t.reshape(B, T, D * H, W))
        tot += m[0].sum().item()
        cnt += y_hat.numel()
    model.train(was_training)
    return tot / max(cnt, 1)


def train_dlwp(model, train_dataset, val_dataset, name="model", epochs=10, batch_size=4, learning_rate=1e-3,
               clip_gradients=False, gradient_accumulation_steps=1, seed=1234, out_dir="outputs", save_model=True,
               use_graph=True, verbose=False, continue_training=False, log_scalars=True):
    """The dlwpbench training script's epoch loop (src/dlwpbench/scripts/train.py:104-197) around the captured step of
    train_engine.GraphedTrainStep, for any dlwpbench module (forward(constants, prescribed, prognostic)): batches of
    `WeatherBenchDataset.__getitem__` tuples (wbdata.WeatherBenchArrays) from the seeded rank-sharded permutation, MSE,
    Adam with the cosine schedule stepped per epoch (:194), optional clipping at max_norm = current learning rate
    (:230-232), micro-batch gradient accumulation (:206-233), validation without gradients, the `_last` / `_best` checkpoint
    policy (:255-266).  One process per GPU; with
    torch.distributed initialised the flat gradient is all-reduced once per step.  Returns per-epoch dicts."""
    from . import wbdata
    from .train_engine import GraphedTrainStep
    # micro-batches as in the reference (:206-233): split_size = max(1, batch // steps); every micro-batch's mean loss is
    # back-propagated un-scaled (the gradients are summed), the clip acts after every micro-backward, one Adam step per batch
    micro = max(1, batch_size // max(1, gradient_accumulation_steps))
    if batch_size % micro != 0:
        raise NotImplementedError("train_dlwp: batch_size must be a multiple of the micro-batch size (static step shapes)")
    device = next(model.parameters()).device
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
    reducer = ddp.FlatGradAllReduce() if world > 1 else None
    ckpt_last = os.path.join(out_dir, name, "checkpoints", f"{name}_last.ckpt")
    step, iteration, best, log = None, 0, float("inf"), []
    epoch0, resume = 0, None
    if continue_training:           # dlwpbench/scripts/train.py:60-71: model, optimizer, epoch, iteration, best error
        resume = torch.load(ckpt_last, map_location="cpu", weights_only=True)
        model.load_state_dict(resume["model_state_dict"])
        epoch0, iteration, best = resume["epoch"], resume["iteration"], resume["best_val_error"]

    def kwargs_of(c, p, g):
        return {k: v for k, v in (("constants", c), ("prescribed", p), ("prognostic", g)) if v is not None}

    writer = ScalarLog(out_dir, name, enabled=log_scalars and save_model and rank == 0)
    for epoch in range(epoch0, epochs):
        lr = cosine_lr(learning_rate, epoch, epochs)
        writer.add_scalar("Epoch", epoch, iteration)
        writer.add_scalar("Learning Rate", lr, iteration)
        losses = []
        if hasattr(train_dataset, "set_epoch"):
            train_dataset.set_epoch(epoch)
        for idx in wbdata.shard_batches(train_dataset, epoch, rank, world, batch_size, seed):
            c, p, g, t = wbdata.to_device_batch([train_dataset[int(i)] for i in idx], device)
            sl = lambda v, i: None if v is None else v[i:i + micro]      # noqa: E731
            if step is None:
                step = GraphedTrainStep(model, kwargs_of(sl(c, 0), sl(p, 0), sl(g, 0)), t[:micro], lr=lr, allreduce=reducer,
                                        grad_scale=1.0 / world, use_graph=use_graph, graph_optimizer=False,
                                        clip_max_norm=lr if clip_gradients else None)
                if resume is not None:
                    st = resume.get("optimizer_state_dict")
                    if not (isinstance(st, dict) and _load_optimizer_state(st, model, step.opt, ckpt_last)):
                        warnings.warn(f"{ckpt_last}: no usable optimizer state; Adam restarts from zero moments")
                    resume = None
                if world > 1:
                    ddp.broadcast_parameters(step.flat, src=0)
                    for buf in (step.opt.exp_avg, step.opt.exp_avg_sq, step.opt.step_count):   # not `t`: that is the target
                        ddp.broadcast_parameters(buf, src=0)
            step.opt.lr = lr
            step.clip = lr if clip_gradients else None
            if micro == batch_size:
                loss = step(kwargs_of(c, p, g), t)
            else:
                for i in range(0, batch_size, micro):
                    loss = step.accumulate(kwargs_of(sl(c, i), sl(p, i), sl(g, i)), t[i:i + micro])
                step.apply()
            losses.append(loss.clone())
            writer.add_scalar("MSE/training", losses[-1], iteration)
            iteration += 1
        train_mse = torch.stack(losses).mean().item() if losses else float("nan")
        val_mse = validation_mse_dlwp(model, val_dataset, batch_size, device)
        writer.add_scalar("MSE/validation", val_mse, iteration)
        writer.flush()
        if save_model and rank == 0 and step is not None:
            sched = {"T_max": epochs, "last_epoch": epoch, "base_lrs": [learning_rate], "_last_lr": [lr]}
            if val_mse > best or epoch == epochs - 1:
                dst = ckpt_last
            else:
                best, dst = val_mse, ckpt_last.replace("last", "best")
            write_checkpoint(model, step.opt, sched, epoch, iteration, best, dst)
        log.append({"epoch": epoch, "lr": lr, "train_mse": train_mse, "val_mse": val_mse})
        if verbose and rank == 0:
            print(f"Epoch {str(epoch).zfill(3)}/{epochs}\tMSE train: {train_mse:.2E}\tMSE val: {val_mse:.2E}")
    return log
