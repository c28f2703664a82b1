#!/usr/bin/env python3
"""Which torch (ATen) kernels still run inside a model's train step, and which source line launches them.

Runs ONE eager forward + backward of a tools/bench_models.py configuration under a TorchDispatchMode that records every ATen
call which launches device work (views and allocations are skipped), keyed by the innermost Python frame inside this package
(calls made by autograd's own C++ nodes -- gradient accumulation at a fork, AddBackward -- have no such frame and are listed
under "<autograd engine>").  A torch.profiler pass of the same step gives the ATen share of the device time.

    python tools/aten_audit.py afno|swin|pangu|...
"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_models  # noqa: E402

NO_KERNEL = ("view", "reshape", "permute", "slice", "select", "transpose", "expand", "as_strided", "detach", "alias", "unsqueeze",
             "squeeze", "empty", "new_empty", "t", "unbind", "split", "chunk", "narrow", "flatten", "unflatten", "is_", "size",
             "stride", "numel", "item", "_local_scalar_dense", "lift_fresh", "movedim", "unfold", "diagonal", "real", "imag",
             "view_as", "_reshape_alias", "_unsafe_view", "sym_size", "sym_stride", "sym_numel", "set_", "resize_", "result_type",
             "record_stream", "empty_like", "empty_strided", "new_empty_strided", "expand_as", "split_with_sizes")


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.calls = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func).replace("aten.", "")
        if name.split(".")[0] in NO_KERNEL:
            return out
        t = out if isinstance(out, torch.Tensor) else next((a for a in args if isinstance(a, torch.Tensor)), None)
        if t is None or not t.is_cuda:
            return out
        where = "<autograd engine>"
        for fr in reversed(traceback.extract_stack()):
            if "dlwp_benchmark_amd/" in fr.filename or fr.filename.endswith("bench_models.py"):
                where = f"{fr.filename.split('dlwp_benchmark_amd/')[-1]}:{fr.lineno} {fr.name}"
                break
        self.calls[(name, where, tuple(t.shape))] += 1
        return out


def audit(name, model, make_batch, steps, warmup=3, use_graph=True, call=None, lr=1e-3):
    dev = torch.device("cuda:0")
    model = model.to(dev).train()
    inputs, target, B = make_batch(dev)
    from dlwp_benchmark_amd.train_engine import flatten_parameters
    flatten_parameters(model)

    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.train_engine import mse_loss, refresh_bf16_weights

    def once():
        # the body of train_engine.GraphedTrainStep._fwd_bwd: bf16 weight copies refreshed and marked live, the engine's loss node
        # (an eager model(**inputs) under bf16 storage takes the per-layer GEMM paths of a model that is not being trained)
        refresh_bf16_weights(model)
        prev, L.SHADOW_ACTIVE = L.SHADOW_ACTIVE, True
        try:
            out = call(model, inputs) if call is not None else model(**inputs)
            loss = mse_loss(out, target)
            loss.backward()
        finally:
            L.SHADOW_ACTIVE = prev
    once()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        once()
        torch.cuda.synchronize()
    total_dev, aten_dev = 0.0, 0.0
    for ev in prof.events():
        if str(ev.device_type).endswith("CUDA"):          # a kernel / memcpy record: counts towards the step's device time
            total_dev += ev.time_range.elapsed_us()
        elif ev.name.startswith("aten::"):
            aten_dev += max(getattr(ev, "self_device_time_total", 0.0), 0.0)
    with Audit() as au:
        once()
    torch.cuda.synchronize()
    n = sum(au.calls.values())
    print(f"== {name}: device time {total_dev / 1e3:.2f} ms, ATen share {aten_dev / max(total_dev, 1e-9) * 100:.1f} %, "
          f"{n} ATen calls with device work")
    by_site = collections.Counter()
    for (op, where, shape), c in au.calls.items():
        by_site[(op, where)] += c
    for (op, where), c in by_site.most_common(60):
        shapes = sorted({s for (o, w, s), _ in au.calls.items() if o == op and w == where})[:3]
        print(f"  {c:5d}x  {op:26s} {where}   {shapes}")


if __name__ == "__main__":
    bench_models.run = audit
    bench_models.main()
