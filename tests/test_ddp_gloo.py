"""World-size-2 gloo test (CPU) of the data-parallel host logic: sharding, flat-bucket all-reduce and
the 1/world scale give the single-process gradient ("2 ranks x b == 1 rank x 2b").
Gradients themselves come from the CPU oracle here; on the GPU box the same reducer is fed by the HIP
trainer (bench.py --gpus N)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dlwp_benchmark_amd import ddp
from oracle import fno_ref

CFG = dict(n_modes=[4, 4], D=1, hidden=8, lifting=16, projection=16, n_layers=2, ctx=2, tf=3, T=5, H=16, W=16)


def _flat_grads(net):
    return torch.cat([torch.view_as_real(p.grad).reshape(-1) if p.is_complex() else p.grad.reshape(-1)
                      for p in net.parameters()])


def _data(n):
    g = torch.Generator().manual_seed(42)
    return torch.randn(n, CFG["T"] + 3, CFG["D"], CFG["H"], CFG["W"], generator=g)


def _grads_on(indices, epoch):
    u = _data(8)
    net = fno_ref.FNO(CFG["n_modes"], CFG["D"] * CFG["ctx"], CFG["hidden"], CFG["lifting"], CFG["projection"],
                      CFG["D"], CFG["n_layers"], seed=7)
    net.requires_grad_(True)
    xs, ys = zip(*[ddp.ns_sample(u, int(i), epoch, CFG["T"] + 1) for i in indices])
    fno_ref.train_step(net, torch.stack(xs), torch.stack(ys), CFG["tf"], CFG["ctx"])
    return _flat_grads(net)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    shard = ddp.shard_indices(8, epoch=3, rank=rank, world=world, batch=2)
    g = _grads_on(shard[0], epoch=3)
    scale = ddp.FlatGradAllReduce()(g)
    if rank == 0:
        torch.save({"grad": g * scale, "shard": shard}, out)
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_with_double_batch(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    shards = [ddp.shard_indices(8, 3, r, 2, 2) for r in range(2)]
    assert np.array_equal(got["shard"], shards[0])
    ref = _grads_on(np.concatenate([shards[0][0], shards[1][0]]), epoch=3)
    err = (got["grad"] - ref).abs().max() / ref.abs().max()
    assert err < 1e-5, err


def test_sharding_is_a_partition_with_equal_iteration_counts():
    for world in (1, 2, 4, 8):
        shards = [ddp.shard_indices(103, epoch=5, rank=r, world=world, batch=3) for r in range(world)]
        assert len({s.shape for s in shards}) == 1                      # same number of iterations everywhere
        flat = np.concatenate([s.reshape(-1) for s in shards])
        assert len(np.unique(flat)) == len(flat)                         # no sample twice
        assert set(flat) <= set(range(103))
    # a sample's crop does not depend on the world size
    assert ddp.crop_start(17, 2, 50, 20) == ddp.crop_start(17, 2, 50, 20)
    a = ddp.epoch_permutation(10, 1)
    assert sorted(a.tolist()) == list(range(10)) and not np.array_equal(a, ddp.epoch_permutation(10, 2))


# ---- dlwp half: WeatherBench batches sharded over two ranks, gradients reduced over gloo -------------------------------------
def _dlwp_grad(items):
    """Flat gradient of a one-step MSE (residual + 1x1 convolution stand-in model, plain torch on the CPU) on dataset items."""
    from dlwp_benchmark_amd import wbdata
    fields, prog, presc, const = wbdata.synthetic_fields(40, 8, 16, prognostic={"t2m": [], "z": [500]}, seed=9)
    ds = wbdata.WeatherBenchArrays(fields, prog, presc, const, sequence_length=2, normalize=True, context_size=1, noise=0.1,
                                   seed=77)
    c, p, g, t = wbdata.to_device_batch([ds[int(i)] for i in items], "cpu")
    gen = torch.Generator().manual_seed(5)
    w = (torch.randn(2, 4 + 1 + 2, generator=gen) * 0.3).requires_grad_(True)      # a stand-in linear "model": 1x1 conv
    x = torch.cat([c[:, 0].expand(len(items), -1, -1, -1), p[:, 0], g[:, 0]], dim=1)          # [B, 7, H, W]
    out = g[:, 0] + torch.einsum("oc,bchw->bohw", w, x)
    loss = torch.nn.functional.mse_loss(out.unsqueeze(1), t)
    loss.backward()
    return w.grad.reshape(-1).clone(), ds


def _dlwp_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from dlwp_benchmark_amd import wbdata
    _, ds = _dlwp_grad([0])
    shard = wbdata.shard_batches(ds, epoch=1, rank=rank, world=world, batch=3)
    g, _ = _dlwp_grad(shard[0])
    scale = ddp.FlatGradAllReduce()(g)
    if rank == 0:
        torch.save({"grad": g * scale, "shard": shard}, out)
    dist.destroy_process_group()


def test_dlwp_batches_two_ranks_equal_one_rank(tmp_path):
    """wbdata sharding + per-item seeded noise + flat all-reduce: 2 ranks x 3 samples == 1 rank x the same 6 samples."""
    from dlwp_benchmark_amd import wbdata
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "r0.pt")
    mp.spawn(_dlwp_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=False)
    _, ds = _dlwp_grad([0])
    shards = [wbdata.shard_batches(ds, 1, r, 2, 3) for r in range(2)]
    assert np.array_equal(got["shard"], shards[0])
    ref, _ = _dlwp_grad(np.concatenate([shards[0][0], shards[1][0]]))
    err = (got["grad"] - ref).abs().max() / ref.abs().max()
    assert err < 1e-5, err


# ---- a failed collective ends the rank with a non-zero exit code ("rank exits non-zero on RCCL error")
def _failing_worker(rank, world, port):
    import datetime
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=20))
    g = torch.ones(1000)
    ddp.FlatGradAllReduceChecked()(g)              # a first, healthy step
    if rank == 1:
        os._exit(0)                                # the peer dies between two steps
    ddp.FlatGradAllReduceChecked()(g)              # rank 0: the collective fails -> os._exit(13)
    os._exit(0)                                    # not reached


def test_collective_error_exits_non_zero():
    ctx = mp.get_context("spawn")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = [ctx.Process(target=_failing_worker, args=(r, 2, port)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
    assert procs[1].exitcode == 0
    assert procs[0].exitcode == 13, procs[0].exitcode


# ---- bucketed reducer: host logic on CPU tensors (world 1: no collective runs, the bookkeeping is what is tested) ------------
def _flat_model(tied=False, deferred=False):
    import torch.nn as nn

    class Leaf(nn.Module):
        def __init__(self, n):
            super().__init__()
            self.w = nn.Parameter(torch.ones(n))

        def forward(self, x):
            return x * self.w.sum()

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.a, self.b, self.c = Leaf(100), Leaf(10), Leaf(10)

        def forward(self, x):
            return self.c(self.b(self.a(x)))

    net = Net()
    if deferred:
        net.b.deferred_grad_writes = True
    flat = torch.zeros(120)
    if tied:      # spans [0,100], [50,60], [70,80]: b and c live INSIDE a's span (shared storage)
        offs = {"a": 0, "b": 50, "c": 70}
    else:
        offs = {"a": 0, "b": 100, "c": 110}
    for name, off in offs.items():
        p = getattr(net, name).w
        p.grad = flat[off:off + p.numel()]
    return net, flat


def test_bucketed_reducer_keeps_only_spans_disjoint_from_kept_ones_and_tiles_the_buffer():
    net, flat = _flat_model(tied=True)
    red = ddp.BucketedGradAllReduce(net, flat, bucket_bytes=1)
    spans = [(b["lo"], b["hi"]) for b in red.buckets]
    assert spans == [(0, 100)], spans                      # [70,80] must not survive because its neighbour [50,60] was dropped
    cover = sorted(spans + red.leftover)
    assert cover[0][0] == 0 and cover[-1][1] == flat.numel() and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))


def test_bucketed_reducer_holds_deferred_buckets_only_and_every_bucket_for_micro_batches():
    x = torch.ones(3, requires_grad=True)
    net, flat = _flat_model()
    red = ddp.BucketedGradAllReduce(net, flat, bucket_bytes=1)
    assert not red.deferred and len(red.buckets) == 3
    net(x).sum().backward()
    assert sum(red._done) >= 2                             # b and c are final before backward reaches a
    red.finish()
    red.hold = True                                        # what GraphedTrainStep.accumulate() sets per micro-batch
    net(x).sum().backward()
    assert sum(red._done) == 0
    red.finish()
    assert red.overlapped == 0 and red.hold is False       # the hold ends with the reduction
    # a module that writes its gradient after its backward hook (deferred_grad_writes) keeps ITS bucket for finish(); the
    # other buckets still leave during backward
    net2, flat2 = _flat_model(deferred=True)
    red2 = ddp.BucketedGradAllReduce(net2, flat2, bucket_bytes=1)
    assert red2.deferred and not red2.hold and red2._bucket_deferred == [False, True, False]
    net2(x).sum().backward()
    assert red2._done == [True, False, True]               # a and c left during backward, b (deferred) waits for finish()
    red2.finish()
    assert red2.overlapped == 2 and not red2.hold


def test_ddp_units_hook_replaces_the_child_walk():
    """A rollout wrapper whose only child is the whole network hands out finer units (ddp_units): they become the buckets."""
    import torch.nn as nn
    net, flat = _flat_model()

    class Wrapper(nn.Module):
        def __init__(self, inner):
            super().__init__()
            self.inner = inner

        def ddp_units(self):
            return [self.inner.a, self.inner.b, self.inner.c]

        def forward(self, x):
            return self.inner(x)
    w = Wrapper(net)
    red = ddp.BucketedGradAllReduce(w, flat, bucket_bytes=1)
    assert [(b["lo"], b["hi"]) for b in red.buckets] == [(0, 100), (100, 110), (110, 120)]
    w(torch.ones(3, requires_grad=True)).sum().backward()
    assert sum(red._done) == 3                             # every unit's backward hook fired on exit (the input needs a gradient)
    red.finish()
