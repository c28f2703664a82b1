"""The data-parallel path on the GPU box: two fresh processes share cuda:0 (gloo rendezvous on 127.0.0.1) and run the HIP
train steps with the gradient all-reduce in between; rank 0 compares with a single-process run of the global batch
(tests/ddp_gpu_worker.py).  Also: the C ABI's own RCCL communicator at world 1 (a one-card box cannot host two RCCL ranks)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_ranks_sharing_the_gpu_equal_one_rank_with_the_global_batch(cuda):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_gpu_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exit {p.returncode}:\n{out[-4000:]}"
    assert "fno trainer" in outs[0] and "bucketed reducer" in outs[0]


def _run_ranks(args, port, world=2, timeout=600):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable] + args, env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exit {p.returncode}:\n{out[-4000:]}"
    return outs


def test_train_dlwp_two_ranks_with_resume_equal_one_rank_with_the_global_batch(cuda, tmp_path):
    """train_loop.train_dlwp end to end at world 2 (parameter + Adam-state broadcast, flat all-reduce, clipping, checkpoint by
    rank 0, continue_training on both ranks) against the uninterrupted single-process run of the global batch."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ddp_gpu_worker as W
    from dlwp_benchmark_amd import train_loop
    from dlwp_benchmark_amd.train_engine import flatten_parameters
    out = str(tmp_path / "two")
    os.makedirs(out)
    _run_ranks([os.path.join(ROOT, "tests", "ddp_gpu_worker.py"), "dlwp", out], 29581)
    got = [torch.load(os.path.join(out, f"rank{r}.pt"), weights_only=False) for r in range(2)]
    assert torch.equal(got[0]["flat"], got[1]["flat"])                 # replicas stay identical
    model, train, val = W.dlwp_case(seed=3)
    model = model.to(cuda)
    log = train_loop.train_dlwp(model, train, val, name="one", epochs=2, batch_size=4, learning_rate=2e-3,
                                out_dir=str(tmp_path), clip_gradients=True)
    ref = flatten_parameters(model)[0].detach().cpu()
    e = ((got[0]["flat"] - ref).abs().max() / ref.abs().max()).item()
    assert e <= 5e-5, e
    assert [r["epoch"] for r in got[0]["log"]] == [1]                  # the resumed run trained epoch 1 only
    assert abs(got[0]["log"][-1]["val_mse"] - log[-1]["val_mse"]) <= 1e-4 * abs(log[-1]["val_mse"])


def test_rccl_communicator_of_the_c_abi_at_world_one(cuda):
    from dlwp_benchmark_amd import ddp
    comm = ddp.RcclComm(rank=0, world=1)
    try:
        g = torch.randn(100003, device=cuda)
        ref = g.clone()
        scale = comm.all_reduce(g)
        comm.broadcast(g, root=0)
        torch.cuda.synchronize()
        assert scale == 1.0 and torch.equal(g, ref)
    finally:
        comm.close()


def test_bucketed_reducer_hooks_release_buckets_during_backward(cuda):
    """World 1: no collective runs, but the hook logic is the same -- every bucket except the one holding the first module
    must be final before backward ends, for a 3-lead-time rollout (modules used three times per step)."""
    from dlwp_benchmark_amd import ddp, nsbench
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    torch.manual_seed(1)
    model = nsbench.AFNONet(img_height=16, img_width=16, patch_size=(2, 2), in_chans=1, out_chans=1, embed_dim=32, depth=4,
                            mlp_ratio=2.0, num_blocks=4, context_size=2).to(cuda).train()
    x = torch.randn(2, 5, 1, 16, 16, device=cuda)
    y = torch.randn(2, 5, 1, 16, 16, device=cuda)
    step = GraphedTrainStep(model, {"x": x}, y, lr=1e-3, use_graph=False, call=lambda m, kw: m(kw["x"], 2))
    red = ddp.BucketedGradAllReduce(model, step.grad, bucket_bytes=4096)
    step.allreduce = red
    covered = sum(b["hi"] - b["lo"] for b in red.buckets) + sum(hi - lo for lo, hi in red.leftover)
    assert covered == step.grad.numel()
    step()
    assert len(red.buckets) >= 4 and red.overlapped >= len(red.buckets) - 1
    with pytest.raises(Exception):
        GraphedTrainStep(model, {"x": x}, y, lr=1e-3, use_graph=True, allreduce=red, call=lambda m, kw: m(kw["x"], 2))


def test_bench_n_greater_than_one_branch_runs_as_two_ranks(cuda):
    """bench.py's own N > 1 code path (rendezvous, per-rank shards, barrier-bracketed timing, MAX over ranks, rank-0 line) as
    two fresh processes that share cuda:0 over gloo -- the driver launches the same file one rank per GPU over RCCL."""
    import json
    outs = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "3",
                       "--warmup", "1", "--no-cpu-baseline", "--no-roofline"], 29583)
    lines = [l for l in outs[0].splitlines() if l.startswith("{")]
    assert len(lines) == 1 and not any(l.startswith("{") for l in outs[1].splitlines())      # ONE line, from rank 0
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak" and line["value"] > 0
    assert line["config"]["global_batch"] == 2 * line["config"]["per_gpu_batch"] and line["config"]["parallelism"] == "dp2"
    assert abs(line["value"] - 2 * line["config"]["per_gpu_batch"] * 3 / (line["ms_per_step"] * 3e-3)) <= 1e-2 * line["value"]
    assert "secondary" not in line and "roofline" not in line
    # rank evidence gathered through the communicator: both ranks, each with its own step time (the line's is their maximum)
    ev = line["ranks"]
    assert ev["world_size"] == 2 and sorted(r["rank"] for r in ev["ranks"]) == [0, 1] and ev["backend"] == "gloo"
    assert max(r["ms_per_step"] for r in ev["ranks"]) <= line["ms_per_step"] * 1.05 + 0.05
    # the autograd-driven models' branch: the step stays a hipGraph at N > 1 (forward + backward | flat all-reduce | optimizer)
    outs = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "2",
                       "--warmup", "1", "--workload", "sfno", "--no-cpu-baseline", "--no-roofline"], 29585)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and line["config"]["hip_graph"] and "flat gradient buffer" in line["config"]["grad_reduce"]
    assert line["value"] > 0 and line["ranks"]["world_size"] == 2
    # ... and the eager variant with the reduction overlapped with backward: SFNO's units are its encoder layers, blocks and
    # decoder layers (ddp_units); the blocks that hold a spectral filter wait for finish() (deferred gradient fold)
    outs = _run_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "2",
                       "--warmup", "1", "--workload", "sfno", "--reduce", "bucketed", "--no-cpu-baseline", "--no-roofline"], 29587)
    line = json.loads([l for l in outs[0].splitlines() if l.startswith("{")][0])
    assert line["n_gpus"] == 2 and not line["config"]["hip_graph"] and line["value"] > 0


def test_collective_captured_inside_the_step_graph_at_world_one(cuda):
    """GraphedTrainStep with the C ABI's RCCL communicator (ddp.RcclComm.in_graph): ncclAllReduce is captured between backward and
    the optimizer; ten replays train exactly like the graph without a collective (at world 1 the sum over ranks is the identity)."""
    import torch
    from dlwp_benchmark_amd import ddp, nsbench
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 1, 16, 16, generator=g).to(cuda)
    y = torch.randn(2, 5, 1, 16, 16, generator=g).to(cuda)
    runs = {}
    for name in ("plain", "in_graph"):
        torch.manual_seed(11)
        m = nsbench.AFNONet(img_height=16, img_width=16, patch_size=(2, 2), in_chans=1, out_chans=1, embed_dim=32, depth=2,
                            mlp_ratio=2.0, num_blocks=4, context_size=2).to(cuda).train()
        comm = ddp.RcclComm(0, 1) if name == "in_graph" else None
        step = GraphedTrainStep(m, {"x": x}, y, lr=1e-3, allreduce=comm, call=lambda mod, kw: mod(kw["x"], 2))
        assert step.collective_in_graph == (name == "in_graph")
        losses = [step().item() for _ in range(10)]
        runs[name] = (losses, step.flat.clone())
        if comm is not None:
            comm.close()
    # (not bit for bit: the weight-gradient float atomics of these small products add in a run-dependent order)
    a, b = torch.tensor(runs["plain"][0]), torch.tensor(runs["in_graph"][0])
    assert a[-1] < a[0] and ((a - b).abs() / a).max().item() < 1e-4
    assert ((runs["plain"][1] - runs["in_graph"][1]).abs().max() / runs["plain"][1].abs().max()).item() < 1e-3
