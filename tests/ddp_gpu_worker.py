"""Worker of tests/test_gpu_ddp.py: one of WORLD_SIZE processes that SHARE cuda:0 (the GPU box has one card), rendezvous over
gloo on 127.0.0.1.  Runs the data-parallel step of the HIP path -- per-rank shard -> captured step -> gradient all-reduce ->
Adam(grad_scale = 1 / world) -- and checks "P ranks x b == 1 rank x P b"  on rank 0 against a single-process
run of the same global batch.  Exit code 0 = every check passed."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from dlwp_benchmark_amd import ddp, nsbench  # noqa: E402
from dlwp_benchmark_amd.train_engine import GraphedTrainStep  # noqa: E402


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def fno(seed):
    torch.manual_seed(seed)
    return nsbench.TFNO2DModule(n_modes=[8, 8], in_channels=1, hidden_channels=16, lifting_channels=32, projection_channels=32,
                                out_channels=1, n_layers=2, context_size=2)


def afno(seed):
    torch.manual_seed(seed)
    return nsbench.AFNONet(img_height=16, img_width=16, patch_size=(2, 2), in_chans=1, out_chans=1, embed_dim=32, depth=3,
                           mlp_ratio=2.0, num_blocks=4, context_size=2)


def dlwp_case(seed=3):
    """Small WeatherBench-shaped datasets + an SFNO2DModule: shared by the 2-rank train_dlwp run below and by the
    single-process run of the global batch in tests/test_gpu_ddp.py."""
    from dlwp_benchmark_amd import dlwpbench, wbdata
    fields, prog, presc, const = wbdata.synthetic_fields(6 * 8 + 8, 16, 32, prognostic={"t2m": [], "z": [500]}, seed=5)
    kw = dict(prognostic_variable_names_and_levels=prog, prescribed_variable_names=presc, constant_names=const,
              sequence_length=4, normalize=True, context_size=1)

    def cut(lo, hi):
        return {k: (v[lo:hi] if not isinstance(v, dict) and v.ndim == 3 else
                    ({l: a[lo:hi] for l, a in v.items()} if isinstance(v, dict) else v)) for k, v in fields.items()}
    train, val = wbdata.WeatherBenchArrays(cut(0, 40), **kw), wbdata.WeatherBenchArrays(cut(40, None), **kw)
    torch.manual_seed(seed)
    model = dlwpbench.SFNO2DModule(constant_channels=4, prescribed_channels=1, prognostic_channels=2, grid="equiangular",
                                   num_layers=2, scale_factor=1, embed_dim=16, context_size=1, height=16, width=32,
                                   big_skip=True, pos_embed=True, use_mlp=True, normalization_layer="none")
    return model, train, val


def dlwp_main(out_dir):
    """train_loop.train_dlwp at world 2 (ranks share cuda:0): one epoch, then continue_training for a second one -- the
    resume path hands rank 0's Adam state to the other rank.  Every rank stores its final parameters; the parent test compares
    them with a single-process run at the global batch."""
    from dlwp_benchmark_amd import train_loop
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    for phase, (epochs, cont) in enumerate(((1, False), (2, True))):
        model, train, val = dlwp_case(seed=3 + rank)           # replicas start different: train_dlwp must broadcast
        model = model.to(dev)
        dist.barrier()                                         # rank 0's checkpoint of the previous phase is on disk
        log = train_loop.train_dlwp(model, train, val, name="w", epochs=epochs, batch_size=2, learning_rate=2e-3,
                                    out_dir=out_dir, continue_training=cont, clip_gradients=True)
        dist.barrier()
    from dlwp_benchmark_amd.train_engine import flatten_parameters
    torch.save({"flat": flatten_parameters(model)[0].detach().cpu(), "log": log}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0)


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "dlwp":
        dlwp_main(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(5)
    u = torch.randn(2 * world, 7, 1, 32, 32, generator=g)            # global batch: 2 samples per rank
    x, y = u[:, :-1].contiguous().to(dev), u[:, 1:].contiguous().to(dev)
    mine = slice(rank, None, world)                                  # rank r takes samples r, r + world, ...
    ok = True

    # ---- (1) fused C++ FNO trainer + one flat all-reduce; replicas start from DIFFERENT seeds and are broadcast
    m = fno(100 + rank).to(dev)
    ddp.broadcast_parameters(m.flat_params.data, src=0)
    opt = m.make_optimizer(lr=1e-3)
    red = ddp.FlatGradAllReduceChecked()
    for _ in range(3):
        m.train_step(x[mine], y[mine], 3, optimizer=opt, grad_scale=1.0 / world, allreduce=red)
    if rank == 0:
        ref = fno(100).to(dev)
        ropt = ref.make_optimizer(lr=1e-3)
        for _ in range(3):
            ref.train_step(x, y, 3, optimizer=ropt)
        e = rel(m.flat_params.data, ref.flat_params.data)
        print(f"fno trainer: params after 3 steps, {world} ranks x 2 vs 1 rank x {2 * world}: rel {e:.2e}", flush=True)
        ok &= e <= 2e-5
    # every rank holds the same parameters
    chk = [torch.zeros(1, device=dev) for _ in range(world)]
    dist.all_gather(chk, m.flat_params.data.double().sum().float().reshape(1))
    ok &= all(abs(c.item() - chk[0].item()) <= 1e-6 * abs(chk[0].item()) for c in chk)

    # ---- (2) autograd-driven model in the captured step: split capture (fwd+bwd | all-reduce | optimizer) and the bucketed
    # reducer launched from backward hooks (eager step); u16: 16 x 16 frames
    g2 = torch.Generator().manual_seed(6)
    u2 = torch.randn(2 * world, 6, 1, 16, 16, generator=g2)
    x2, y2 = u2[:, :-1].contiguous().to(dev), u2[:, 1:].contiguous().to(dev)
    call = lambda mm, kw: mm(kw["x"], 2)   # noqa: E731
    results = {}
    for mode in ("flat", "bucketed"):
        model = afno(7).to(dev).train()
        if mode == "flat":
            step = GraphedTrainStep(model, {"x": x2[mine]}, y2[mine], lr=1e-3, allreduce=ddp.FlatGradAllReduceChecked(),
                                    grad_scale=1.0 / world, use_graph=True, call=call)
        else:
            step = GraphedTrainStep(model, {"x": x2[mine]}, y2[mine], lr=1e-3, grad_scale=1.0 / world, use_graph=False, call=call)
            step.allreduce = ddp.BucketedGradAllReduce(model, step.grad, bucket_bytes=4096)
        for _ in range(3):
            step()
        results[mode] = step.flat.clone()
        if mode == "bucketed":
            nb, ov = len(step.allreduce.buckets), step.allreduce.overlapped
            if rank == 0:
                print(f"bucketed reducer: {nb} buckets, {ov} released during backward", flush=True)
            ok &= nb >= 3 and ov >= nb - 1
    if rank == 0:
        ref = afno(7).to(dev).train()
        rstep = GraphedTrainStep(ref, {"x": x2}, y2, lr=1e-3, use_graph=True, call=call)
        for _ in range(3):
            rstep()
        for mode, flat in results.items():
            e = rel(flat, rstep.flat)
            print(f"afno captured step ({mode} reduce): rel {e:.2e}", flush=True)
            ok &= e <= 5e-5
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
