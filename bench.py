#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the nsbench FNO autoregressive rollout step on
MI355X (BASELINE.json metric; workload = configs[1]: TFNO2DModule 64x64 Navier-Stokes,
T=20 frames, context 10, teacher forcing 10 => 11 net calls, 10 of them closed loop, fp32).

One step = rollout forward + MSE + BPTT backward (one hipGraph) + gradient all-reduce
(N>1, RCCL) + fused Adam, on one batch of synthetic trajectories already resident in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
WORKLOAD = dict(name="nsbench TFNO2DModule 64x64 NS rollout (BASELINE configs[1])",
                n_modes=[12, 12], in_channels=1, hidden_channels=32, lifting_channels=256,
                projection_channels=256, out_channels=1, n_layers=4, context_size=10,
                T=20, teacher_forcing_steps=10, H=64, W=64)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4, help="per-GPU batch (reference: training.batch_size=4)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def pwmlp_bwd_flops(B, HW, Cin, Ch, Cout):
    """algorithmic FLOPs of one projection/lifting backward launch:
    recompute W1 x, W2^T gy, W1^T gz, gy.act^T, gz.x^T -> 2*P*Ch*(3*Cin + 2*Cout)"""
    return 2.0 * B * HW * Ch * (3 * Cin + 2 * Cout)


def roofline_probe(device, B, reps=200):
    """Time the dominant kernel (projection backward: pwmlp_bwd<2,1>, the same instantiation
    and launch geometry the captured step uses) with HIP events on the launch stream."""
    import torch
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    w = WORKLOAD
    HW, Cin, Ch, Cout = w["H"] * w["W"], w["hidden_channels"], w["projection_channels"], w["out_channels"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, HW, generator=g).to(device)
    w1 = (torch.randn(Ch, Cin, generator=g) / Cin ** 0.5).to(device)
    b1 = torch.zeros(Ch, device=device)
    w2 = (torch.randn(Cout, Ch, generator=g) / Ch ** 0.5).to(device)
    gy = torch.randn(B, Cout, HW, generator=g).to(device)
    gx = torch.empty_like(x)
    gw1, gb1, gw2, gb2 = torch.zeros_like(w1), torch.zeros_like(b1), torch.zeros_like(w2), torch.zeros(Cout, device=device)
    stream = torch.cuda.current_stream()

    def launch():
        L.check(lib.dlwp_pwmlp_bwd(L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(gy), L.ptr(gx), L.ptr(gw1),
                                   L.ptr(gb1), L.ptr(gw2), L.ptr(gb2), B, Cin, Ch, Cout, HW, stream.cuda_stream))
    for _ in range(20):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        launch()
    e1.record(stream)
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    flops = pwmlp_bwd_flops(B, HW, Cin, Ch, Cout)
    achieved = flops / sec / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("pwmlp_bwd_kernel<2,1>")
        except Exception:
            traffic = None
    return {"bound": "mfma", "kernel": "pwmlp_bwd_kernel<2,1> (projection backward)",
            "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "flops_per_launch": flops,
            "us_per_launch": round(sec * 1e6, 3), "traffic": traffic}


def cpu_baseline(B, budget_s):
    """The oracle (CPU restatement of the reference path: neuralop is not installable here, so
    kind="port") timed on the host cores: same workload, same step (fwd + MSE + BPTT + Adam)."""
    import torch
    from oracle import fno_ref
    w = WORKLOAD
    torch.set_num_threads(os.cpu_count() or 1)
    net = fno_ref.FNO(w["n_modes"], w["in_channels"] * w["context_size"], w["hidden_channels"],
                      w["lifting_channels"], w["projection_channels"], w["out_channels"], w["n_layers"], seed=1234)
    net.requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    u = torch.randn(B, w["T"] + 1, 1, w["H"], w["W"], generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    fno_ref.train_step(net, x, y, w["teacher_forcing_steps"], w["context_size"], optimizer=opt)  # warm-up
    t0 = time.perf_counter()
    n = 0
    while True:
        fno_ref.train_step(net, x, y, w["teacher_forcing_steps"], w["context_size"], optimizer=opt)
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"{n} train steps of the same workload (batch {B}, T={w['T']}) after 1 warm-up, "
            f"torch {torch.__version__} CPU fp32, oracle/fno_ref.py"}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from dlwp_benchmark_amd import nsbench

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product has no CPU path)")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    w = WORKLOAD
    B = args.batch
    torch.manual_seed(1234)
    model = nsbench.TFNO2DModule(n_modes=w["n_modes"], in_channels=w["in_channels"],
                                 hidden_channels=w["hidden_channels"], lifting_channels=w["lifting_channels"],
                                 projection_channels=w["projection_channels"], out_channels=w["out_channels"],
                                 n_layers=w["n_layers"], context_size=w["context_size"]).to(device)
    if world > 1:
        dist.broadcast(model.flat_params.data, src=0)
    opt = model.make_optimizer(lr=1e-3)
    # synthetic trajectories (seeded per rank: every rank trains on its own shard), resident in HBM
    g = torch.Generator().manual_seed(1234 + rank)
    u = torch.randn(B, w["T"] + 1, 1, w["H"], w["W"], generator=g).to(device)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    allreduce = (lambda gbuf: dist.all_reduce(gbuf, op=dist.ReduceOp.SUM)) if world > 1 else None
    scale = 1.0 / world

    def step():
        return model.train_step(x, y, w["teacher_forcing_steps"], optimizer=opt, use_graph=not args.no_graph,
                                grad_scale=scale, allreduce=allreduce)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    final_loss = loss.item()

    if rank == 0:
        ncalls = w["T"] - w["context_size"] + 1
        line = {
            "metric": "train samples/sec (FNO 64x64 rollout step: fwd + MSE + BPTT + Adam)",
            "value": round(world * B * args.steps / dt, 3), "unit": "samples/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic N(0,1) trajectories, random-init weights (no dataset/checkpoint access)",
            "config": {"workload": w["name"], "per_gpu_batch": B, "global_batch": B * world, "T": w["T"],
                       "context_size": w["context_size"], "teacher_forcing_steps": w["teacher_forcing_steps"],
                       "net_calls_per_sample": ncalls, "hidden_channels": w["hidden_channels"],
                       "n_layers": w["n_layers"], "n_modes": w["n_modes"], "grid": [w["H"], w["W"]],
                       "parallelism": f"dp{world}", "hip_graph": not args.no_graph},
            "backbone_calls_per_s": round(world * B * args.steps * ncalls / dt, 1),
            "final_loss": final_loss,
        }
        if world == 1 and not args.no_roofline:
            line["roofline"] = roofline_probe(device, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
