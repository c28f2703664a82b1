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
