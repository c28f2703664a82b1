#!/usr/bin/env python3
"""Who launches the bf16 cast kernels (token_ops._lowp / _scaled_grad) in one eager training step of a bench.py workload: call sites
with shapes.    python tools/cast_census.py swin|pangu|afno721"""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from dlwp_benchmark_amd import dlwpbench, lib as L, token_ops as TO  # noqa: E402
from dlwp_benchmark_amd.train_engine import GraphedTrainStep  # noqa: E402

wl = sys.argv[1]
w = bench.DLWP_WORKLOADS[wl]
L.set_gemm_precision("bf16")
L.set_storage(w["storage"])
B, H, W_, Cg, T, lr = w["batch"], w["H"], w["W"], w["Cg"], w["T"], w.get("lr", 1e-3)
dev = torch.device("cuda", 0)
torch.manual_seed(1234)
model = getattr(dlwpbench, w["cls"])(**w["model"]).to(dev).train()
g = torch.Generator().manual_seed(1234)
kw = dict(constants=torch.randn(B, 1, 4, H, W_, generator=g).to(dev), prescribed=torch.randn(B, T, 1, H, W_, generator=g).to(dev),
          prognostic=torch.randn(B, T, Cg, H, W_, generator=g).to(dev))
target = torch.randn(B, T - 1, Cg, H, W_, generator=g).to(dev)
step = GraphedTrainStep(model, kw, target, lr=lr, clip_max_norm=lr, use_graph=False)
step._fwd_bwd()
census = collections.Counter()


def wrap(name):
    orig = getattr(TO, name)

    def f(t, *a, **k):
        out = orig(t, *a, **k)
        if t is not None and out is not t:
            fr = [x for x in traceback.extract_stack()[:-1] if "dlwp_benchmark_amd" in x.filename][-2:]
            census[(name, tuple(t.shape), " <- ".join(f"{os.path.basename(x.filename)}:{x.lineno} {x.name}" for x in reversed(fr)))] += 1
        return out
    setattr(TO, name, f)


TO._GRAD_LOWP["hits"] = TO._GRAD_LOWP["misses"] = 0
wrap("_lowp")
wrap("_scaled_grad")
step._fwd_bwd()
torch.cuda.synchronize()
print("LayerNorm-backward bf16 copies: hits", TO._GRAD_LOWP["hits"], "misses", TO._GRAD_LOWP["misses"])
for (name, shape, where), n in sorted(census.items(), key=lambda kv: -kv[1] * (kv[0][1][0] if kv[0][1] else 1)):
    print(f"{n:3d}x {name:13s} {str(shape):18s} {where}")
