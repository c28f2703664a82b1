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
