"""Data-parallel host logic of the rollout training path (a build addition: the
reference's train.py is single-process, nsbench/scripts/train.py:36,66).

One process per GPU.  Units = trajectory samples (no cross-sample coupling anywhere on the path), so
the only exchange is ONE sum all-reduce of the flat gradient buffer per optimizer step (RCCL over xGMI
through torch.distributed's "nccl" backend on the GPU box, "gloo" in the CPU tests), followed by
Adam with grad_scale = 1/world.  MSELoss is a mean over equal-sized shards, so the mean of the per-rank
gradients equals the single-process gradient of the global batch (up to fp32 summation order).
"""
import numpy as np
import torch
import torch.distributed as dist


def epoch_permutation(n_samples, epoch, seed=1234):
    """The one seeded permutation every rank derives independently (reference seed: configs/config.yaml:13)."""
    rng = np.random.default_rng([seed, epoch])
    return rng.permutation(n_samples)


def shard_indices(n_samples, epoch, rank, world, batch, seed=1234, drop_last=True):
    """Indices this rank trains on in `epoch`, as a [n_iters, batch] array.

    rank r takes perm[r::world]; the tail is dropped (or wrapped when drop_last=False) so that every
    rank runs the same number of iterations — a rank that ran fewer would dead-lock the all-reduce."""
    perm = epoch_permutation(n_samples, epoch, seed)
    per_rank = len(perm) // world if drop_last else -(-len(perm) // world)
    if not drop_last and per_rank * world > len(perm):
        perm = np.concatenate([perm, perm[: per_rank * world - len(perm)]])
    mine = perm[rank::world][:per_rank]
    n_iters = len(mine) // batch
    return mine[: n_iters * batch].reshape(n_iters, batch)


def crop_start(index, epoch, t_file, length, seed=1234):
    """NavierStokesDataset.__getitem__ random crop start (nsbench/data/datasets/datasets.py:40), seeded
    per (epoch, sample) so the data a sample contributes does not depend on the number of ranks."""
    rng = np.random.default_rng([seed, epoch, int(index)])
    return int(rng.integers(0, t_file - length + 1))


def ns_sample(u, index, epoch, length, noise=0.0, seed=1234):
    """x = u[i, r:r+L-1] (+ N(0, noise^2)), y = u[i, r+1:r+L]  (datasets.py:39-44); u: [N,T,D,H,W] tensor."""
    r = crop_start(index, epoch, u.shape[1], length, seed)
    x = u[index, r:r + length - 1].clone()
    y = u[index, r + 1:r + length].clone()
    if noise > 0.0:
        g = torch.Generator().manual_seed(int(np.random.default_rng([seed, epoch, int(index), 7]).integers(2 ** 31)))
        x = x + noise * torch.randn(x.shape, generator=g)
    return x, y


class FlatGradAllReduce:
    """Sum-all-reduce of the flat gradient bucket; returns the scale Adam must apply (1/world)."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, flat_grad):
        if self.world > 1:
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world


def broadcast_parameters(flat_params, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)


def _fail(what, exc):
    """A failed collective leaves the replicas inconsistent: report on stderr and end this rank with a non-zero exit code
    (no elastic recovery on this path; the launcher -- torchrun -- then tears the other ranks down)."""
    import os
    import sys
    rank = dist.get_rank() if dist.is_initialized() else 0
    print(f"[dlwp_benchmark_amd.ddp] rank {rank}: {what} failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
    sys.stderr.flush()
    os._exit(13)


class FlatGradAllReduceChecked(FlatGradAllReduce):
    """FlatGradAllReduce that turns a collective error (peer died, timeout, RCCL failure) into exit code 13."""

    def __call__(self, flat_grad):
        try:
            return super().__call__(flat_grad)
        except Exception as exc:          # noqa: BLE001 -- torch raises RuntimeError / DistBackendError / DistNetworkError
            _fail("gradient all-reduce", exc)


class BucketedGradAllReduce:
    """Gradient all-reduce in buckets launched DURING backward (north_star: "all-reduce of gradients over xGMI overlapped
    with backward"), for the autograd-driven models whose parameters live in one flat buffer
    (train_engine.flatten_parameters): AFNONet / SwinTransformer / PanguWeather at the 1-lead-time C4 / C5 configurations
    carry 114 - 288 MB of fp32 gradients, and the last layers' gradients are final long before the first layers'.

    Units = the sub-modules of `model` (children of ModuleLists expanded); a unit's parameters occupy one contiguous span of
    the flat gradient buffer.  A unit is FINAL once the backward pass has left it as often as the forward pass entered it
    (a module used T times in a rollout: T forward pre-hooks, T full-backward hooks); a hook that fires without input
    gradients (a module fed by data only) fires on entry, not on exit, and is ignored.  Consecutive spans are grouped into
    buckets of >= `bucket_bytes`; a bucket is reduced as soon as ALL its units are final -- no assumption about the order in
    which autograd visits the modules.  Reductions are asynchronous (`async_op=True`: RCCL's stream waits for the kernels
    enqueued so far, later backward kernels overlap).  `finish()` reduces everything no hook released (the first unit, gaps
    between spans such as position embeddings owned by the parent), waits for all reductions and resets the counters.
    The FNO rollout modules keep ONE bucket (FlatGradAllReduce): BPTT accumulates every parameter gradient over all net
    calls, so nothing is final before backward ends.
    """

    def __init__(self, model, flat_grad, bucket_bytes=25 * 2 ** 20, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.flat_grad = flat_grad
        base, total = flat_grad.data_ptr(), flat_grad.numel()
        spans = []                                    # (lo, hi, module): element range of the unit's parameters
        for unit in self._units(model):
            ps = [p for p in unit.parameters() if p.requires_grad and p.grad is not None]
            if not ps:
                continue
            lo = min((p.grad.data_ptr() - base) // 4 for p in ps)
            hi = max((p.grad.data_ptr() - base) // 4 + p.numel() for p in ps)
            if 0 <= lo and hi <= total and sum(p.numel() for p in ps) + 8 * len(ps) >= hi - lo:     # contiguous up to padding
                spans.append((lo, hi, unit))
        # a block that batches its weight gradients (token_ops.WgradBatch) writes them when its FIRST layer runs backward: its
        # sub-modules' backward hooks fire before that, so no unit may be a proper part of such a block
        unit_ids = {id(sp[2]) for sp in spans}
        for blk in model.modules():
            if getattr(blk, "wgrad_batch_block", False):
                inner = [type(m).__name__ for m in blk.modules() if m is not blk and id(m) in unit_ids]
                if inner:
                    raise RuntimeError(f"BucketedGradAllReduce: units {inner} lie inside a {type(blk).__name__} whose weight gradients are "
                                       "written together at the end of the block's backward: a unit must be the block or larger")
        spans.sort(key=lambda t: t[0])
        kept, end = [], 0                             # drop units that overlap an already KEPT span (tied / shared parameters)
        for sp in spans:
            if sp[0] >= end:
                kept.append(sp)
                end = sp[1]
        spans = kept
        self.buckets = []                             # dict(lo, hi, units)
        for lo, hi, unit in spans:
            last = self.buckets[-1] if self.buckets else None
            if last and (last["hi"] - last["lo"]) * 4 < bucket_bytes and lo - last["hi"] <= 8:
                last["hi"] = hi
                last["units"].append(unit)
            else:
                self.buckets.append({"lo": lo, "hi": hi, "units": [unit]})
        self.leftover, pos = [], 0                    # ranges outside every bucket: reduced by finish()
        for bk in self.buckets:
            if bk["lo"] > pos:
                self.leftover.append((pos, bk["lo"]))
            pos = bk["hi"]
        if pos < total:
            self.leftover.append((pos, total))
        cover = sorted([(bk["lo"], bk["hi"]) for bk in self.buckets] + self.leftover)
        assert all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and (not cover or (cover[0][0] == 0 and cover[-1][1] == total)), \
            "buckets + leftover must tile the flat gradient buffer exactly once"
        # Gradients must be FINAL when a unit's backward hook fires.  Two paths break that: modules that defer a gradient
        # write to the end of the backward pass (class attribute `deferred_grad_writes`: a gradient folded by an engine
        # callback) -- only the buckets that CONTAIN such a module wait for finish() -- and micro-batch accumulation (later
        # micro-batches add local gradients on top of an already reduced sum): `hold` keeps every bucket back and finish()
        # reduces the whole buffer once.
        self._bucket_deferred = [any(getattr(m, "deferred_grad_writes", False) for u in bk["units"] for m in u.modules())
                                 for bk in self.buckets]
        self.deferred = any(self._bucket_deferred)
        self.hold = False
        self._unit_bucket, self._fwd, self._bwd = {}, {}, {}
        self._handles, self._works = [], []
        for j, bk in enumerate(self.buckets):
            for unit in bk["units"]:
                self._unit_bucket[id(unit)] = j
                self._fwd[id(unit)] = self._bwd[id(unit)] = 0
                self._handles.append(unit.register_forward_pre_hook(self._on_forward))
                self._handles.append(unit.register_full_backward_hook(self._on_backward))
        self._done = [False] * len(self.buckets)

    @staticmethod
    def _units(model):
        """Direct children, ModuleLists / Sequentials expanded; a module may name its own units (`ddp_units()`: e.g. the rollout
        wrapper SFNO2DModule, whose only child is the whole network, hands out the network's encoder layers, blocks and decoder
        layers), which are then used as they are."""
        import torch.nn as nn
        if hasattr(model, "ddp_units"):
            return list(model.ddp_units())
        out = []
        for child in model.children():
            if isinstance(child, (nn.ModuleList, nn.Sequential)) or hasattr(child, "ddp_units"):
                out.extend(BucketedGradAllReduce._units(child))
            else:
                out.append(child)
        return out

    def _on_forward(self, module, args):
        import torch
        if torch.is_grad_enabled():
            self._fwd[id(module)] += 1

    def _on_backward(self, module, grad_input, grad_output):
        if all(g is None for g in grad_input):
            return                                    # fired on ENTRY (no input needs a gradient): parameters not final yet
        k = id(module)
        self._bwd[k] += 1
        j = self._unit_bucket[k]
        units = self.buckets[j]["units"]        # a unit the forward pass never entered (fwd == 0) receives no gradient at all
        if not self.hold and not self._bucket_deferred[j] and not self._done[j] and all(self._bwd[id(u)] >= self._fwd[id(u)] for u in units):
            self._launch(self.buckets[j]["lo"], self.buckets[j]["hi"])
            self._done[j] = True

    def _launch(self, lo, hi):
        if self.world == 1 or hi <= lo:
            return
        try:
            self._works.append(dist.all_reduce(self.flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        except Exception as exc:          # noqa: BLE001
            _fail("bucketed gradient all-reduce", exc)

    def finish(self):
        """Reduce what no hook released, wait for every reduction, reset the counters.  Returns Adam's scale 1 / world."""
        self.overlapped = sum(self._done)             # buckets that went out during backward (diagnostics / tests)
        self.hold = False                             # a micro-batch hold ends with the reduction
        for j, bk in enumerate(self.buckets):
            if not self._done[j]:
                self._launch(bk["lo"], bk["hi"])
        for lo, hi in self.leftover:
            self._launch(lo, hi)
        try:
            for w in self._works:
                w.wait()
        except Exception as exc:          # noqa: BLE001
            _fail("bucketed gradient all-reduce", exc)
        self._works = []
        self._done = [False] * len(self.buckets)
        for k in self._fwd:
            self._fwd[k] = self._bwd[k] = 0
        return 1.0 / self.world

    def __call__(self, flat_grad=None):
        return self.finish()

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class RcclComm:
    """The C ABI's own RCCL communicator (include/dlwpmi.h: dlwp_comm_*), for hosts that do not route the exchange through
    torch.distributed.  The 128-byte unique id travels over whatever side channel the host has; here: an already initialised
    torch.distributed group of ANY backend (gloo is enough) or, at world 1, nothing at all.
    `in_graph = True`: train_engine.GraphedTrainStep captures the all-reduce inside the step's hipGraph (the call only enqueues
    ncclAllReduce on the current stream)."""

    in_graph = True

    def __init__(self, rank=0, world=1, group=None):
        import ctypes as C
        from . import lib as L
        self.L, self.lib = L, L.load()
        uid = (C.c_char * 128)()
        if rank == 0:
            L.check(self.lib.dlwp_comm_unique_id(C.cast(uid, C.c_void_p)))
        if world > 1:
            box = [bytes(uid)]
            dist.broadcast_object_list(box, src=0, group=group)
            uid = (C.c_char * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        L.check(self.lib.dlwp_comm_create(C.cast(uid, C.c_void_p), rank, world, C.byref(h)))
        self.h, self.rank, self.world = h, rank, world

    def all_reduce(self, flat):
        self.L.check(self.lib.dlwp_comm_allreduce(self.h, self.L.ptr(flat), flat.numel(), self.L.stream()))
        return 1.0 / self.world

    __call__ = all_reduce

    def broadcast(self, flat, root=0):
        self.L.check(self.lib.dlwp_comm_broadcast(self.h, self.L.ptr(flat), flat.numel(), root, self.L.stream()))

    def close(self):
        if self.h:
            self.lib.dlwp_comm_destroy(self.h)
            self.h = None
