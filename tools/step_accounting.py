#!/usr/bin/env python3
"""Live per-kernel accounting of one workload's training step with GEMM rows split by product shape (dlwp_prof_enable(2)).

    python tools/step_accounting.py pangu|swin|afno|afno721|sfno [reps] [rows]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from dlwp_benchmark_amd import lib as L  # noqa: E402


def main():
    wl = sys.argv[1]
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    from dlwp_benchmark_amd import dlwpbench
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    w = bench.DLWP_WORKLOADS[wl]
    L.set_gemm_precision("bf16")
    L.set_storage(w["storage"])
    B, H, W_, Cg, T, lr = w["batch"], w["H"], w["W"], w["Cg"], w["T"], w.get("lr", 1e-3)
    device = torch.device("cuda", 0)
    torch.manual_seed(1234)
    model = getattr(dlwpbench, w["cls"])(**w["model"]).to(device).train()
    g = torch.Generator().manual_seed(1234)
    kw = dict(constants=torch.randn(B, 1, 4, H, W_, generator=g).to(device), prescribed=torch.randn(B, T, 1, H, W_, generator=g).to(device),
              prognostic=torch.randn(B, T, Cg, H, W_, generator=g).to(device))
    target = torch.randn(B, T - 1, Cg, H, W_, generator=g).to(device)
    step = GraphedTrainStep(model, kw, target, lr=lr, clip_max_norm=lr, use_graph=False)
    info = f"B={B} storage={w['storage']}"
    for _ in range(2):
        step._fwd_bwd()
        step._optimize()
    torch.cuda.synchronize()
    with L.kernel_accounting(shapes=True) as acc:
        for _ in range(reps):
            step._fwd_bwd()
            step._optimize()
        torch.cuda.synchronize()
    tot = sum(r["ms"] for r in acc.rows)
    print(f"{wl}: accounted {tot / reps:.3f} ms/step over {len(acc.rows)} rows  ({info})")
    for r in acc.rows[:nrows]:
        us = r["ms"] * 1e3 / r["calls"]
        print(f"  {r['name'][:100]:100s} {100 * r['ms'] / tot:5.1f} %  x{r['calls'] / reps:5.1f}  {us:7.1f} us  "
              f"{r['flops'] / (r['ms'] * 1e-3) / 1e12 if r['ms'] else 0:7.1f} TF  {r['bytes'] / (r['ms'] * 1e-3) / 1e9 if r['ms'] else 0:7.0f} GB/s")


if __name__ == "__main__":
    main()
