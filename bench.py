#!/usr/bin/env python3
"""Headline benchmark: train samples/sec of the nsbench FNO autoregressive rollout step on
MI355X (BASELINE.json metric; workload = configs[1]: TFNO2DModule 64x64 Navier-Stokes,
T=20 frames, context 10, teacher forcing 10 => 11 net calls, 10 of them closed loop, fp32).

One step = rollout forward + MSE + BPTT backward (one hipGraph) + gradient all-reduce
(N>1, RCCL) + fused Adam, on one batch of synthetic trajectories already resident in HBM.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (no launcher: starts the command above itself, see spawn_ranks)

The N = 1 line carries, after the headline: "secondary" = configs[2] (SFNO) on the reference's dlwpbench protocol and "tertiary" =
configs[3] (Swin, Pangu at 128x256, window 7) and configs[4] (FourCastNet AFNO at 721x1440), each with its own value, roofline
(the kernel that leads that workload's step, measured live) and cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
T_PROCESS_START = time.time()
# rough wall time of one CPU leg of the tertiary list (warm-up step + one timed step, 8 - 16 host threads): legs that do not fit what
# is left of --time-budget are skipped, never the GPU measurements
CPU_LEG_SECONDS = {"swin": 25, "pangu": 60, "afno721": 80}

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
    ap.add_argument("--hidden", type=int, default=None, help="hidden_channels of the fno workload (default 32 = the headline; "
                    "the published sweep runs 2 ... 217, src/nsbench/scripts/train_commands.txt:83-91)")
    ap.add_argument("--T", type=int, default=None, help="frames per trajectory of the fno workload (default 20 = the headline; the "
                    "reference's own protocol is sequence_length 50 -> --T 49: 40 net calls per sample, "
                    "src/nsbench/configs/training/default.yaml:6, scripts/train_commands.txt:83)")
    ap.add_argument("--no-secondary", action="store_true", help="fno workload at N=1: do not append the short SFNO (configs[2]) run "
                    "as the line's \"secondary\" object")
    ap.add_argument("--no-tertiary", action="store_true", help="fno workload at N=1: do not append the configs[3] / [4] runs (Swin, Pangu, "
                    "AFNO 721x1440) as the line's \"tertiary\" list")
    ap.add_argument("--time-budget", type=float, default=420.0, help="wall seconds the default N = 1 run may take (from process start): the "
                    "tertiary workloads and their CPU legs are skipped once it is nearly used up (the driver stops bench.py at 600 s)")
    ap.add_argument("--no-spawn", action="store_true", help="--gpus N > 1 without a launcher (WORLD_SIZE unset): exit with an error that "
                    "names the torch.distributed.run command instead of starting the N ranks")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--split-graph", action="store_true", help="N = 1 only: run the step with the graph structure of the N > 1 path "
                    "(dlwpbench workloads: forward + backward graph | reducer call | optimizer graph, the reducer being a no-op at world "
                    "1) -- what the data-parallel structure costs before any byte moves.  The fno workload's step has this structure "
                    "at every N (forward + backward graph | reducer | Adam launch), so the flag changes nothing there")
    ap.add_argument("--no-clip", action="store_true", help="dlwpbench workloads: no gradient clipping (the reference's training protocol "
                    "clips at max_norm = learning rate before every optimizer step: src/dlwpbench/configs/training/default.yaml:3, "
                    "scripts/train.py:133-135; on by default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the "
                    "single-GPU functional test of the N>1 code path)")
    ap.add_argument("--share-device", action="store_true", help="all ranks use cuda:0 (functional test only)")
    ap.add_argument("--reduce", default="auto", choices=["auto", "flat", "in-graph", "bucketed"],
                    help="gradient reduction of the sfno / pangu / swin / afno workloads at N > 1.  flat: the step stays a hipGraph -- "
                         "forward + backward | ONE all-reduce of the flat gradient buffer (torch.distributed, RCCL) | optimizer; "
                         "in-graph: the same with the all-reduce captured inside the graph through the C ABI's own RCCL communicator "
                         "(dlwp_comm_allreduce); bucketed: eager step, buckets all-reduced from backward hooks while backward runs "
                         "(overlap instead of capture).  auto (default) = by measurement on one card (profiles/r04_nograph_lines.jsonl): "
                         "sfno loses 41 %% of its rate without the graph -> flat; pangu / swin / afno lose <= 1.5 %% -> bucketed")
    ap.add_argument("--workload", default="fno", choices=["fno", "sfno", "pangu", "swin", "afno", "afno721"],
                    help="fno: BASELINE configs[1] (default, the headline line); sfno: configs[2], dlwpbench SFNO2DModule 32x64, "
                         "5 prognostic variables, sfno.yaml widths, sequence length 5 (4 lead times); pangu / swin: configs[3] "
                         "(128x256, window 7); afno: configs[4] grid (FourCastNet 720x1440, patch 8, E=768, depth 12); afno721: the "
                         "same network on the 721x1440 grid BASELINE names, patch (7, 8) -> 103 x 180 tokens")
    ap.add_argument("--storage", default=None, choices=["fp32", "bf16"],
                    help="sfno workload: storage of GEMM-to-GEMM activations and of the weight copy the GEMMs read (default: bf16 "
                         "with bf16 operands)")
    ap.add_argument("--precision", default=None, choices=["fp32", "bf16"],
                    help="GEMM operand precision of the sfno workload (default bf16 = the reference's autocast, fp32 accumulate)")
    a = ap.parse_args()
    a.batch_given = any(x == "--batch" or x.startswith("--batch=") for x in sys.argv[1:])
    return a


def launcher_command(args_list, n, port="P"):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(args_list)


def spawn_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset) must not quietly benchmark ONE GPU and print
    "n_gpus": 1.  This process has not imported torch or touched a GPU yet: it starts the documented torch.distributed.run
    command as a CHILD (never an exec), relays the children's output -- rank 0 prints the JSON line -- and exits with the
    child's code.  --no-spawn: fail with the command in the message instead."""
    import socket
    import subprocess
    argv = [a for a in sys.argv[1:] if a != "--no-spawn"]
    if args.no_spawn:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is not set: this process would run ONE rank.  Launch the ranks with\n  "
                         + " ".join(launcher_command(argv, args.gpus)) + "\n(or drop --no-spawn and bench.py starts them itself)")
    with socket.socket() as sk:          # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launcher_command(argv, args.gpus, port)
    print("bench.py: WORLD_SIZE unset with --gpus %d: starting %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
PEAK_MFMA_TF = {"fp32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md dense peaks


def spatial_fwd_bytes(B, C, H, W, m1, m2c):
    """algorithmic HBM bytes of one forward `spatial` launch (kernel table): read the
    input field, write the pre-activation field, read the mixed modes, write the next block's row
    spectrum, read skip weights + bias."""
    field = 4.0 * B * C * H * W
    return 2 * field + 8.0 * B * m1 * m2c * C + 8.0 * B * H * m2c * C + 4.0 * (C * C + C)


def roofline_probe(device, B, reps=300):
    """Time the dominant kernel of the step (rocprof: fno_spatial_kernel<2,1>, ~35% of GPU time; same
    instantiation, grid and fused stages as inside the captured step) with HIP events on its stream."""
    import ctypes as C_
    import torch
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    w = WORKLOAD
    C, H, W = w["hidden_channels"], w["H"], w["W"]
    m1, m2c = w["n_modes"][0], w["n_modes"][1] // 2 + 1
    plan = C_.c_void_p()
    L.check(lib.dlwp_fno_plan_create(C, H, W, m1, m2c, C_.byref(plan)))
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, C, H, W, generator=g).to(device)
    spec = (torch.randn(B, m1, m2c, C, 2, generator=g) * 0.1).to(device)
    wskip = (torch.randn(C, C, generator=g) / C ** 0.5).to(device)
    bias = torch.zeros(C, device=device)
    pre = torch.empty_like(x)
    x1 = torch.empty(B, H, m2c, C, 2, device=device)
    stream = torch.cuda.Stream()

    def launch():
        L.check(lib.dlwp_fno_spatial_fwd_probe(plan, L.ptr(x), L.ptr(spec), L.ptr(wskip), L.ptr(bias), L.ptr(pre),
                                               L.ptr(x1), B, stream.cuda_stream))
    with torch.cuda.stream(stream):
        for _ in range(20):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            launch()
        e1.record(stream)
        torch.cuda.synchronize()
    lib.dlwp_fno_plan_destroy(plan)
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    nbytes = spatial_fwd_bytes(B, C, H, W, m1, m2c)
    achieved = nbytes / sec / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("fno_spatial_kernel<2,1>")
        except Exception:
            traffic = None
    kname = "fno_spatial_kernel<2,1> (forward, inner block)" if C <= 64 else \
        "fno_spatial_wide_kernel<0,1> + fno_rows_wide_kernel (forward, inner block)"
    return {"bound": "hbm", "kernel": kname,
            "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(achieved / PEAK_HBM_GBS, 4), "bytes_per_launch": nbytes,
            "us_per_launch": round(sec * 1e6, 3), "traffic": traffic}


def mfma_probe(device, B, reps=200):
    """Second roofline figure (north_star: MFMA utilisation next to the HBM figure): the step's matrix-core kernels are the
    lifting / projection MLPs (37 % of GPU time).  Times the lifting backward (`pwmlp_bwd_kernel<1,2>`, slab mode as in the
    step) with HIP events on its launch stream; FLOPs = 2 P Ch (3 Cin + 2 Cout): z, g_a, dW2, dW1, dX products, padding and
    the in-register transposition not counted."""
    import torch
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    w = WORKLOAD
    Cin, Ch, Cout, P = w["in_channels"] * w["context_size"], w["lifting_channels"], w["hidden_channels"], w["H"] * w["W"]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, P, generator=g).to(device)
    w1 = (torch.randn(Ch, Cin, generator=g) / Cin ** 0.5).to(device)
    b1 = torch.zeros(Ch, device=device)
    w2 = (torch.randn(Cout, Ch, generator=g) / Ch ** 0.5).to(device)
    gy = torch.randn(B, Cout, P, generator=g).to(device)
    gx = torch.empty_like(x)
    slab = torch.zeros(lib.dlwp_pwmlp_slab_floats(B, Cin, Ch, Cout, P), device=device)
    stream = torch.cuda.Stream()

    def launch():
        L.check(lib.dlwp_pwmlp_bwd_slab(L.ptr(x), L.ptr(w1), L.ptr(b1), L.ptr(w2), L.ptr(gy), L.ptr(gx), L.ptr(slab), 1, B, Cin,
                                        Ch, Cout, P, stream.cuda_stream))
    with torch.cuda.stream(stream):
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
    flops = 2.0 * B * P * Ch * (3 * Cin + 2 * Cout)
    return {"bound": "mfma", "kernel": "pwmlp_bwd_kernel<1,2> (lifting MLP backward, slab mode)", "achieved": round(flops / sec / 1e12, 2),
            "peak": PEAK_MFMA_TF["fp32"], "unit": "TFLOP/s", "frac": round(flops / sec / 1e12 / PEAK_MFMA_TF["fp32"], 4),
            "flops_per_launch": flops, "us_per_launch": round(sec * 1e6, 3)}


def mix_probe(device, B, reps=300):
    """Third roofline figure (north_star: MFMA utilisation on the spectral contraction): one forward per-mode launch
    (fno_mix_fwd_kernel: H-axis step + the mode-truncated complex weight contraction on v_mfma_f32_16x16x4_f32), timed with
    HIP events on its launch stream.  FLOPs: contraction 8 B C^2 per mode + H-axis step 8 B C H per mode; bytes: the x1
    slices (read once per kept column in the ideal case), the weights, xhat and y."""
    import ctypes as C_
    import torch
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    w = WORKLOAD
    C, H, W = w["hidden_channels"], w["H"], w["W"]
    m1, m2c = w["n_modes"][0], w["n_modes"][1] // 2 + 1
    plan = C_.c_void_p()
    L.check(lib.dlwp_fno_plan_create(C, H, W, m1, m2c, C_.byref(plan)))
    g = torch.Generator().manual_seed(0)
    x1 = torch.randn(B, H, m2c, C, 2, generator=g).to(device)
    wspec = (torch.randn(m1, m2c, C, C, 2, generator=g) / C ** 0.5).to(device)
    xhat = torch.empty(B, m1, m2c, C, 2, device=device)
    y = torch.empty(B, m1, m2c, C, 2, device=device)
    stream = torch.cuda.Stream()

    def launch():
        L.check(lib.dlwp_fno_mix_fwd_probe(plan, L.ptr(x1), L.ptr(wspec), L.ptr(xhat), L.ptr(y), B, stream.cuda_stream))
    with torch.cuda.stream(stream):
        for _ in range(20):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            launch()
        e1.record(stream)
        torch.cuda.synchronize()
    lib.dlwp_fno_plan_destroy(plan)
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    modes = m1 * m2c
    flops = 8.0 * B * C * C * modes + 8.0 * B * C * H * modes
    nbytes = 8.0 * (B * H * m2c * C + modes * C * C + 2 * B * modes * C)
    return {"bound": "mfma", "kernel": "fno_mix_fwd_kernel (H-axis step + per-mode complex contraction on MFMA)",
            "achieved": round(flops / sec / 1e12, 3), "peak": PEAK_MFMA_TF["fp32"], "unit": "TFLOP/s",
            "frac": round(flops / sec / 1e12 / PEAK_MFMA_TF["fp32"], 5), "flops_per_launch": flops,
            "bytes_per_launch": nbytes, "GBps": round(nbytes / sec / 1e9, 1), "us_per_launch": round(sec * 1e6, 3)}


def cpu_baseline(B, budget_s):
    """The oracle (CPU restatement of the reference path: neuralop is not installable here, so
    kind="port") timed on the host cores: same workload, same step (fwd + MSE + BPTT + Adam)."""
    import torch
    from oracle import fno_ref
    w = WORKLOAD
    # more threads than this slow the small FNO ops down (measured on the 256-core GPU host: 8 -> 9.8,
    # 16 -> 10.8, 32 -> 4.7, 64 -> 1.5 samples/s)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    net = fno_ref.FNO(w["n_modes"], w["in_channels"] * w["context_size"], w["hidden_channels"],
                      w["lifting_channels"], w["projection_channels"], w["out_channels"], w["n_layers"], seed=1234)
    net.requires_grad_(True)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    u = torch.randn(B, w["T"] + 1, 1, w["H"], w["W"], generator=g)
    x, y = u[:, :-1].contiguous(), u[:, 1:].contiguous()
    # protocol of SURVEY.md §8(d): 3 warm-up + >= 10 timed iterations of forward + backward + Adam, MEDIAN step time
    for _ in range(3):
        fno_ref.train_step(net, x, y, w["teacher_forcing_steps"], w["context_size"], optimizer=opt)
    times = []
    t_start = time.perf_counter()
    while len(times) < 10 or (time.perf_counter() - t_start < budget_s and len(times) < 40):
        t0 = time.perf_counter()
        fno_ref.train_step(net, x, y, w["teacher_forcing_steps"], w["context_size"], optimizer=opt)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(B / med, 3), "unit": "samples/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": f"median of {len(times)} train steps of the same workload (batch {B}, T={w['T']}) after 3 warm-ups, "
            f"torch {torch.__version__} CPU fp32, oracle/fno_ref.py (the reference's FNO needs neuralop: not installable here)"}


SFNO_WORKLOAD = dict(name="dlwpbench SFNO2DModule 32x64 WeatherBench shapes (BASELINE configs[2])",
                     model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=5, grid="equiangular", num_layers=4,
                                scale_factor=1, embed_dim=256, context_size=1, height=32, width=64, big_skip=True, pos_embed=True,
                                use_mlp=True, normalization_layer="none"), T=5, H=32, W=64)


# BASELINE configs[3] and [4] as further --workload choices (supplementary lines: the headline is configs[1]).  model kwargs are the
# shipped dlwpbench YAMLs at the grids BASELINE names; one lead time per step (sequence length 2), as the dlwpbench training runs.
DLWP_WORKLOADS = {
    # per-GPU batch 16 = the reference's own batch_size (src/dlwpbench/configs/training/default.yaml:4); --batch 4 --no-clip is the
    # round 1 - 4 line
    "sfno": dict(cls="SFNO2DModule", name=SFNO_WORKLOAD["name"], model=SFNO_WORKLOAD["model"], T=5, H=32, W=64, Cg=5, batch=16,
                 storage="bf16", gemm=(32 * 64, 512, 256), metric="train samples/sec (SFNO 32x64 rollout step: fwd + MSE + backward + Adam)"),
    "pangu": dict(cls="PanguWeather", name="dlwpbench PanguWeather 128x256 window (2,7,7) (BASELINE configs[3])",
                  model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=8, embed_dim=192, num_heads=(6, 12, 12, 6),
                             window_size=(2, 7, 7), patch_size=(1, 1), n_lat=128, n_lon=256, context_size=1),
                  T=2, H=128, W=256, Cg=8, batch=1, storage="bf16", gemm=(128 * 256, 768, 192), lr=1e-4,
                  metric="train samples/sec (Pangu-Weather 128x256 window-7 step: fwd + MSE + backward + Adam)"),
    "swin": dict(cls="SwinTransformer", name="dlwpbench SwinTransformer 128x256 window 7 (BASELINE configs[3])",
                 model=dict(constant_channels=4, prescribed_channels=1, prognostic_channels=8, context_size=1, img_height=128,
                            img_width=256, patch_size=1, embed_dim=96, depths=[4, 4], num_heads=[4, 4], drop_path_rate=0.2,
                            window_size=7),
                 T=2, H=128, W=256, Cg=8, batch=2, storage="bf16", gemm=(128 * 256, 384, 96),      # (round 5: 6.47 vs 6.69 ms with fp32 storage)
                 metric="train samples/sec (Swin 128x256 window-7 step: fwd + MSE + backward + Adam)"),
    "afno": dict(cls="AFNONet", name="dlwpbench AFNONet (FourCastNet) 720x1440 patch 8 E768 depth 12 (BASELINE configs[4] grid)",
                 model=dict(img_height=720, img_width=1440, patch_size=(8, 8), constant_channels=4, prescribed_channels=1,
                            prognostic_channels=8, embed_dim=768, depth=12, mlp_ratio=4.0, num_blocks=16, context_size=1),
                 T=2, H=720, W=1440, Cg=8, batch=1, storage="bf16", gemm=(90 * 180, 3072, 768),
                 metric="train samples/sec (FourCastNet AFNO 720x1440 step: fwd + MSE + backward + Adam)"),
    "afno721": dict(cls="AFNONet", name="dlwpbench AFNONet (FourCastNet) 721x1440 patch (7,8) E768 depth 12 (BASELINE configs[4])",
                    model=dict(img_height=721, img_width=1440, patch_size=(7, 8), constant_channels=4, prescribed_channels=1,
                               prognostic_channels=8, embed_dim=768, depth=12, mlp_ratio=4.0, num_blocks=16, context_size=1),
                    T=2, H=721, W=1440, Cg=8, batch=1, storage="bf16", gemm=(103 * 180, 3072, 768),
                    metric="train samples/sec (FourCastNet AFNO 721x1440 step: fwd + MSE + backward + Adam)"),
}


def sfno_gemm_probe(device, B, precision, reps=100, storage="fp32", shape=(32 * 64, 512, 256)):
    """Dominant kernel family of the token models (rocprof: gemm_kernel): the block MLP's first layer [B*tokens, K] x [K, N]
    with bias + GELU epilogue (SFNO: 2048 tokens, 256 -> 512), timed with HIP events on its launch stream."""
    import torch
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _gemm
    M, N, K = B * shape[0], shape[1], shape[2]
    g = torch.Generator().manual_seed(0)
    x = torch.randn(M, K, generator=g).to(device)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(device)
    b = torch.zeros(N, device=device)
    y, z = torch.empty(M, N, device=device), torch.empty(M, N, device=device)
    if storage == "bf16":         # as in the step: LayerNorm output, weight copy, hidden activation and pre-activation are bf16 arrays
        x, w, y, z = x.to(torch.bfloat16), w.to(torch.bfloat16), y.to(torch.bfloat16), z.to(torch.bfloat16)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(10):
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1, b, 1, z, None)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            _gemm(x, w, y, M, N, K, K, K, N, 0, 1, b, 1, z, None)
        e1.record(stream)
        torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    flops = 2.0 * M * N * K
    peak = PEAK_MFMA_TF[precision]
    # both roofs: a shallow product (SFNO: K = 256) moves more bytes than it has flops for -- algorithmic bytes = x and W read once,
    # the activated output and the stored pre-activation written once
    nbytes = float(x.element_size() * M * K + w.element_size() * N * K + (y.element_size() + z.element_size()) * M * N + 4 * N)
    f_mfma, f_hbm = flops / sec / 1e12 / peak, nbytes / sec / 1e9 / PEAK_HBM_GBS
    out = {"bound": "mfma" if f_mfma >= f_hbm else "hbm",
           "kernel": f"block-MLP fc1 GEMM {M}x{N}x{K} with the bias + GELU + stored pre-activation epilogue, {precision} operands, {storage} "
                     f"storage of x / W / h / z (gemm_glds_kernel when both operands are bf16 arrays, K >= 256 and >= 256 output tiles; "
                     f"gemm_kernel otherwise)",
           "flops_per_launch": flops, "bytes_per_launch": nbytes, "us_per_launch": round(sec * 1e6, 3), "traffic": None,
           "mfma": {"achieved": round(flops / sec / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(f_mfma, 4)},
           "hbm": {"achieved": round(nbytes / sec / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}}
    out.update(out[out["bound"]])          # the contract's achieved / peak / unit / frac = the roof that binds this shape
    return out


def _time_on_stream(launch, reps):
    """Average duration of `launch(stream_handle)` over `reps` back-to-back calls, HIP events on the launch stream."""
    import torch
    torch.cuda.synchronize()
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        for _ in range(10):
            launch(stream.cuda_stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(reps):
            launch(stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


def _both_roofs(kernel, flops, nbytes, sec, traffic=None):
    peak = PEAK_MFMA_TF["bf16"]
    f_mfma, f_hbm = flops / sec / 1e12 / peak, nbytes / sec / 1e9 / PEAK_HBM_GBS
    out = {"bound": "mfma" if f_mfma >= f_hbm else "hbm", "kernel": kernel, "flops_per_launch": flops, "bytes_per_launch": nbytes,
           "us_per_launch": round(sec * 1e6, 3), "traffic": traffic,
           "mfma": {"achieved": round(flops / sec / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(f_mfma, 4)},
           "hbm": {"achieved": round(nbytes / sec / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(f_hbm, 4)}}
    out.update(out[out["bound"]])          # the contract's achieved / peak / unit / frac = the roof that binds
    return out


def sfno_tail_probe(device, B, reps=100, tokens_per_sample=32 * 64, C=256, hidden=512, backward=False):
    """The SFNO block tail (rocprof: mlp_chain_kernel, forward + backward 25 - 32 % of the step): one launch of dlwp_sfno_tail_fwd
    (inner skip + GELU + fc1 + GELU + fc2 + outer skip over B * 2048 tokens) or of dlwp_sfno_tail_bwd, timed with HIP events on its
    launch stream.  Algorithmic bytes forward = x, y read and out written in fp32, the five bf16 tensors kept for the backward
    pass (x copy, GELU'(z0), t, GELU'(z1), h) written once, the three weight matrices read once; backward = g read, gt and gx
    written in fp32, the two stored derivatives read, g / gh / gt bf16 copies written; flops = the three products."""
    import ctypes as C_
    import torch
    from dlwp_benchmark_amd import lib as L
    from dlwp_benchmark_amd.token_ops import _TailBwdArgs, _TailFwdArgs
    lib = L.load()
    T, bf = B * tokens_per_sample, torch.bfloat16
    g = torch.Generator().manual_seed(0)
    x, y = torch.randn(T, C, generator=g).to(device), torch.randn(T, C, generator=g).to(device)
    ws, w1, w2 = ((torch.randn(r, c, generator=g) / c ** 0.5).to(device) for r, c in ((C, C), (hidden, C), (C, hidden)))
    bs, b1, b2 = torch.zeros(C, device=device), torch.zeros(hidden, device=device), torch.zeros(C, device=device)
    imgs = torch.empty(6, C * hidden, device=device, dtype=bf)
    L.check(lib.dlwp_sfno_tail_pack(L.ptr(ws), L.ptr(w1), L.ptr(w2), C, hidden, L.ptr(imgs), L.stream()))
    e = lambda n, dt=bf: torch.empty(T, n, device=device, dtype=dt)
    x_lp, z0, t, z1, h, out = e(C), e(C), e(C), e(hidden), e(hidden), e(C, torch.float32)
    fa = _TailFwdArgs(L.ptr(x), L.ptr(y), L.ptr(imgs[0]), L.ptr(imgs[1]), L.ptr(imgs[2]), L.ptr(bs), L.ptr(b1), L.ptr(b2), L.ptr(x_lp),
                      L.ptr(z0), L.ptr(t), L.ptr(z1), L.ptr(h), L.ptr(out), T, C, hidden, 1)
    L.check(lib.dlwp_sfno_tail_fwd(C_.byref(fa), L.stream()))          # (the backward probe reads the derivatives this call stores)
    flops = 2.0 * T * (C * C + 2 * C * hidden)
    wbytes = 2 * (C * C + 2 * C * hidden)
    if backward:
        g_lp, gh, gt, gt_lp, gx = e(C), e(hidden), e(C, torch.float32), e(C), e(C, torch.float32)
        ba = _TailBwdArgs(L.ptr(y), L.ptr(imgs[3]), L.ptr(imgs[4]), L.ptr(imgs[5]), L.ptr(z1), L.ptr(z0), L.ptr(g_lp), L.ptr(gh),
                          L.ptr(gt), L.ptr(gt_lp), L.ptr(gx), T, C, hidden, 1)
        sec = _time_on_stream(lambda st: L.check(lib.dlwp_sfno_tail_bwd(C_.byref(ba), st)), reps)
        nbytes = float(T * C * (4 + 4 + 4) + T * (C + hidden) * 2 + T * (2 * C + hidden) * 2 + wbytes)
        name = f"mlp_chain_kernel<{C}, {hidden}, {C}, {C}, 2, true> (dlwp_sfno_tail_bwd)"
    else:
        sec = _time_on_stream(lambda st: L.check(lib.dlwp_sfno_tail_fwd(C_.byref(fa), st)), reps)
        nbytes = float(T * C * (4 + 4 + 4) + T * C * 2 * 3 + T * hidden * 2 * 2 + wbytes + 4 * (2 * C + hidden))
        name = f"mlp_chain_kernel<{C}, {C}, {hidden}, {C}, 2, false> (dlwp_sfno_tail_fwd)"
    return _both_roofs(f"{name}: the SFNO block tail's three dependent products in one launch over {T} tokens, bf16 operands, "
                       f"fp32 accumulation", flops, nbytes, sec)


def sfno_spectral_probe(device, B, kind, reps=100, K=32, N=64, C=256, M=32, Lm=32):
    """The three spectral kernels of an SFNO block at the C3 shape (32 x 64 grid, lmax = mmax = 32, embed 256), one launch each
    between HIP events on the launch stream.  kind: "synthesis" (sht_synthesis_bf16_kernel, truncated orders not read; the backward
    half of its calls also adds a residual field: not in this probe), "analysis" (sht_analysis_bf16_kernel), "dhconv"
    (dhconv_apply2_kernel, truncated).  Algorithmic bytes: the spectrum entries with m <= l (bf16, 33 / 64 of the array; dhconv
    also writes the zero rows of its live 8-order tiles: counted as the array's 5 / 8) and the field read or written once in
    fp32, the weights / tables once; flops = the two table products resp. the complex channel mixing on the live rows."""
    import torch
    from dlwp_benchmark_amd import lib as L
    lib = L.load()
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    tri = (torch.arange(M)[None, :] <= torch.arange(Lm)[:, None]).float()
    X = (torch.randn(Lm, B, M, 2, C, generator=g) * tri[:, None, :, None, None]).to(device).to(bf).contiguous()
    live = float(tri.sum()) / (Lm * M)
    spec_bytes = 2.0 * Lm * B * M * 2 * C
    field_bytes = 4.0 * B * K * N * C
    if kind == "dhconv":
        img = (torch.randn(Lm * 2 * C * C, generator=g) / 16).to(device).to(bf)
        Y = torch.empty_like(X)
        sec = _time_on_stream(lambda st: L.check(lib.dlwp_dhconv_apply(L.ptr(X), L.ptr(img), L.ptr(Y), Lm, B * M * 2, C, C, M, 0, st)), reps)
        tiles = sum(min(M // 8, l // 8 + 1) for l in range(Lm)) / (Lm * (M // 8))          # live 8-order tiles
        flops = 8.0 * C * C * B * M * Lm * tiles
        nbytes = spec_bytes * tiles + spec_bytes + 2.0 * Lm * 2 * C * C
        return _both_roofs(f"dhconv_apply2_kernel<{C}, 128> (dlwp_dhconv_apply): per-degree complex channel mixing of {B * M * 2} spectrum "
                           f"rows x {Lm} degrees, bf16 operands, truncated orders skipped", flops, nbytes, sec)
    if kind == "analysis":
        x = torch.randn(B, K, N, C, generator=g).to(device)
        A1 = (torch.randn(2 * M, N, generator=g) / 8).to(device).to(bf)
        A2 = (torch.randn(M, Lm, K, generator=g) / 8).to(device).to(bf)
        sec = _time_on_stream(lambda st: L.check(lib.dlwp_sht_analysis_bf16(L.ptr(x), L.ptr(A1), L.ptr(A2), L.ptr(X), B, K, N, C, M, Lm, st)),
                              reps)
        flops = 2.0 * B * C * K * N * 2 * M + 2.0 * B * C * 2 * M * Lm * K
        return _both_roofs(f"sht_analysis_bf16_kernel<{(N + 31) // 32}, {(K + 31) // 32}, {(Lm + 15) // 16}> (dlwp_sht_analysis_bf16): "
                           f"longitude DFT + Legendre transform of {B} x {C} fields {K} x {N}", flops, field_bytes + spec_bytes, sec)
    out = torch.empty(B, K, N, C, device=device)
    S1t = (torch.randn(M, K, Lm, generator=g) / 8).to(device).to(bf)
    S2 = (torch.randn(N, 2 * M, generator=g) / 8).to(device).to(bf)
    sec = _time_on_stream(lambda st: L.check(lib.dlwp_sht_synthesis_bf16_ex(L.ptr(X), L.ptr(S1t), L.ptr(S2), None, L.ptr(out), B, K, N, C,
                                                                            M, Lm, 1, st)), reps)
    flops = 2.0 * B * C * K * 2 * M * Lm * live + 2.0 * B * C * K * N * 2 * M
    return _both_roofs(f"sht_synthesis_bf16_kernel<{(Lm + 31) // 32}, {(2 * M + 31) // 32}, {(N + 15) // 16}> (dlwp_sht_synthesis_bf16_ex): "
                       f"Legendre synthesis + longitude inverse DFT of {B} x {C} fields {K} x {N}, truncated orders not read",
                       flops, spec_bytes * live + field_bytes, sec)


SFNO_PROFILE_CSV = {16: ["r06_bf16_storage_sfno_b16_step_kernel_stats.csv", "r05_bf16_storage_sfno_b16_step_kernel_stats.csv"],
                    4: ["r06_bf16_storage_sfno_step_kernel_stats.csv", "r05_bf16_storage_sfno_step_kernel_stats.csv"]}          # newest first


def sfno_dominant_probe(device, B):
    """secondary.roofline: the kernel with the largest share of the step's committed rocprof kernel-stats CSV (profiles/), probed
    live.  Falls back to the block tail's forward launch when the CSV of this batch size is absent."""
    import csv
    share, top = None, "mlp_chain_kernel<256, 256, 512, 256, 2, false>"
    path = ""
    for name in SFNO_PROFILE_CSV.get(B, []):
        if os.path.isfile(os.path.join(ROOT, "profiles", name)):
            path = os.path.join(ROOT, "profiles", name)
            break
    if os.path.isfile(path):
        rows = list(csv.DictReader(open(path)))
        if rows:
            top, share = rows[0]["Name"], float(rows[0]["Percentage"])
    if "sht_synthesis" in top:
        r = sfno_spectral_probe(device, B, "synthesis")
    elif "sht_analysis" in top:
        r = sfno_spectral_probe(device, B, "analysis")
    elif "dhconv_apply" in top:
        r = sfno_spectral_probe(device, B, "dhconv")
    else:
        r = sfno_tail_probe(device, B, backward=", true>" in top)
    r["share_of_step_pct"] = share
    r["chosen_from"] = os.path.relpath(path, ROOT) if share is not None else "default (no committed CSV for this batch size)"
    return r


# committed rocprofv3 step tables (profiles/README.md) per workload, newest first
STEP_TABLE_CSV = {
    "swin": ["r06_bf16_storage_swin_c4_step_kernel_stats.csv", "r05_bf16_storage_swin_c4_step_kernel_stats.csv"],
    "pangu": ["r06_bf16_storage_pangu_c4_step_kernel_stats.csv", "r05_bf16_storage_pangu_c4_step_kernel_stats.csv"],
    "afno": ["r06_bf16_storage_afno_fcn_step_kernel_stats.csv", "r05_bf16_storage_afno_fcn_step_kernel_stats.csv"],
    "afno721": ["r06_bf16_storage_afno721_step_kernel_stats.csv"],
}


def kernel_short_name(rocprof_name):
    """rocprofv3's kernel name without `void`, the anonymous namespace and the argument list:
    'void (anonymous namespace)::gemm_glds_kernel<false, 32>((anonymous namespace)::GemmDev)' -> 'gemm_glds_kernel<false, 32>'."""
    n = rocprof_name.replace("(anonymous namespace)::", "").strip()
    if n.startswith("void "):
        n = n[5:]
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):            # the argument list opens at the first '(' outside the template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    return n[:cut].strip()


def committed_step_table(workload):
    """(relative path, [(short kernel name, share %, average us)]) of the newest committed step table of this workload, or (None, [])."""
    import csv
    for name in STEP_TABLE_CSV.get(workload, []):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.isfile(path):
            rows = [(kernel_short_name(r["Name"]), float(r["Percentage"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(path))]
            if rows:
                return os.path.relpath(path, ROOT), rows
    return None, []


def live_dominant_roofline(step, workload, reps=2):
    """roofline of the kernel that leads this workload's step.  `reps` EAGER steps (the captured step's own body: same launches,
    shapes and epilogues) run under the library's live accounting (csrc/prof.hip: an event pair around every instrumented launch
    on its own stream, algorithmic flops / bytes from the launch arguments).  The kernel is the one that leads the committed
    rocprofv3 table of the workload (profiles/r06_*_step_kernel_stats.csv) when the live table has it, else the live table's own
    leader; achieved = its summed algorithmic work / its summed event time."""
    import torch
    from dlwp_benchmark_amd import lib as L
    step._fwd_bwd()          # one eager step outside the accounting: allocator warm-up of the eager path
    step._optimize()
    torch.cuda.synchronize()
    with L.kernel_accounting() as acc:
        for _ in range(reps):
            step._fwd_bwd()
            step._optimize()
        torch.cuda.synchronize()
    rows = acc.rows
    if not rows:
        return None
    total_ms = sum(r["ms"] for r in rows)
    path, table = committed_step_table(workload)
    pick, share, rocprof_us, why = None, None, None, None
    if table:
        top_name, share, rocprof_us = table[0]
        for r in rows:
            if r["name"] == top_name or r["name"].split("<")[0] == top_name.split("<")[0] and "<" not in r["name"]:
                pick, why = r, f"leads {path}"
                break
    if pick is None:
        pick = rows[0]
        why = ("leads the live table" + (f" ({path} is led by {table[0][0]}, which the live accounting does not cover)" if table else
                                          " (no committed step table for this workload yet)"))
        hit = [t for t in table if t[0] == pick["name"]]
        share, rocprof_us = (hit[0][1], hit[0][2]) if hit else (None, None)
    sec = pick["ms"] * 1e-3 / pick["calls"]
    out = _both_roofs(pick["name"], pick["flops"] / pick["calls"], pick["bytes"] / pick["calls"], sec)
    out.update({"calls_per_step": pick["calls"] / reps, "share_of_step_pct": share, "chosen_from": path, "chosen_because": why,
                "rocprof_avg_us": None if rocprof_us is None else round(rocprof_us, 2),
                "live_share_of_accounted_pct": round(100.0 * pick["ms"] / total_ms, 2),
                "measured": f"HIP-event pair around each of the kernel's {pick['calls']} launches in {reps} eager steps of this workload "
                            "(dlwp_prof_*), flops / bytes per launch averaged over those launches; an event pair adds ~2 us per launch",
                "live_table": [{"kernel": r["name"], "calls_per_step": r["calls"] / reps, "us_per_launch": round(r["ms"] * 1e3 / r["calls"], 2),
                                "pct": round(100.0 * r["ms"] / total_ms, 2),
                                "TFLOPs": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 2) if r["ms"] > 0 else None,
                                "GBs": round(r["bytes"] / (r["ms"] * 1e-3) / 1e9, 1) if r["ms"] > 0 else None} for r in rows[:8]]})
    return out


def sfno_cpu_baseline(B, budget_s, clip=None):
    """oracle/sfno_ref.py (CPU restatement; torch-harmonics is not installable here: kind="port") on the host cores; clip: the
    reference protocol's torch.nn.utils.clip_grad_norm_ threshold (dlwpbench scripts/train.py:133-135)."""
    import torch
    from oracle import sfno_ref
    from dlwp_benchmark_amd import dlwpbench
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    cfg = SFNO_WORKLOAD["model"]
    torch.manual_seed(1234)
    net = dlwpbench.SFNO2DModule(**cfg)          # parameter container only (CPU tensors); the arithmetic is the oracle's
    p = {k[len("sfno."):]: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    opt = torch.optim.Adam(list(p.values()), lr=1e-3)
    g = torch.Generator().manual_seed(1234)
    T = SFNO_WORKLOAD["T"]
    kw = (torch.randn(B, 1, 4, 32, 64, generator=g), torch.randn(B, T, 1, 32, 64, generator=g), torch.randn(B, T, 5, 32, 64, generator=g))
    target = torch.randn(B, T - 1, 5, 32, 64, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(sfno_ref.sfno2d_rollout(*kw, p, cfg), target)
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(list(p.values()), clip)
        opt.step()
    step()
    t0, n = time.perf_counter(), 0
    while True:
        step()
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 50:
            break
    dt = time.perf_counter() - t0
    return {"value": round(B * n / dt, 3), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n} train steps of the same workload (batch {B}, {T - 1} lead times{', clip_grad_norm_' if clip is not None else ''}) "
                      f"after 1 warm-up, torch {torch.__version__} CPU fp32, oracle/sfno_ref.py"}


def dlwp_cpu_baseline(workload, B, budget_s, clip=None, threads=None):
    """cpu_baseline of the configs[3] / [4] lines: one training step (rollout + MSE + backward + clip + Adam) of the CPU oracle
    (oracle/{swin,pangu,afno}_ref.py: restatements pinned to the reference's own classes by tests/golden/*.npz; kind="port") at
    the benchmarked shape, fp32, on the host cores.  A step of these 28 - 74 M parameter models takes seconds to a minute on a
    CPU, so the sample is bounded: ONE warm-up step, then steps are timed until `budget_s` is used up, at least one.  The reference's own classes timed in the build container are in profiles/r06_cpu_reference_c4_c5.json
    (tools/cpu_reference_c4_c5.py)."""
    import ctypes
    import torch
    from dlwp_benchmark_amd import dlwpbench
    w = DLWP_WORKLOADS[workload]
    B = B or w["batch"]
    torch.set_num_threads(threads or min(os.cpu_count() or 1, 16))
    try:          # multi-GB activations allocated and freed every step: keep them in the heap (M_MMAP_THRESHOLD, M_TRIM_THRESHOLD,
        libc = ctypes.CDLL("libc.so.6")          # M_TOP_PAD) instead of mmap / munmap + page faults per tensor -- in the CPU's favour
        libc.mallopt(-3, 1 << 30), libc.mallopt(-1, (1 << 31) - 1), libc.mallopt(-2, 1 << 28)
    except OSError:
        pass
    cfg = dict(w["model"])
    torch.manual_seed(1234)
    net = getattr(dlwpbench, w["cls"])(**cfg)          # parameter container only (CPU tensors); the arithmetic is the oracle's
    p = {k: v.detach().clone() for k, v in net.state_dict().items()}
    names = {n for n, _ in net.named_parameters()}
    for k in p:
        if k in names:
            p[k].requires_grad_(True)
    leaves = [p[k] for k in p if k in names]
    del net
    opt = torch.optim.Adam(leaves, lr=w.get("lr", 1e-3))
    if workload == "swin":
        from oracle import swin_ref
        cfg.setdefault("patch_norm", True)
        fwd = lambda c, pr, pg: swin_ref.dlwp_swin(c, pr, pg, p, cfg)            # noqa: E731  (drop_path: training-mode stochastic depth is
        note = "oracle/swin_ref.py (dlwp_swin, window 7; stochastic depth not applied: every block always runs)"       # not restated)
    elif workload == "pangu":
        from oracle import pangu_ref
        fwd = lambda c, pr, pg: pangu_ref.rollout(c, pr, pg, p, cfg)            # noqa: E731
        note = "oracle/pangu_ref.py (rollout; stochastic depth not applied)"
    else:
        from oracle import afno_ref
        fwd = lambda c, pr, pg: afno_ref.dlwp_afnonet(c, pr, pg, p, cfg)        # noqa: E731
        note = "oracle/afno_ref.py (dlwp_afnonet)"
    H, W_, Cg, T = w["H"], w["W"], w["Cg"], w["T"]
    g = torch.Generator().manual_seed(1234)
    c, pr, pg = (torch.randn(B, 1, 4, H, W_, generator=g), torch.randn(B, T, 1, H, W_, generator=g),
                 torch.randn(B, T, Cg, H, W_, generator=g))
    target = torch.randn(B, T - 1, Cg, H, W_, generator=g)

    def step():
        opt.zero_grad(set_to_none=True)
        loss = torch.nn.functional.mse_loss(fwd(c, pr, pg), target)
        loss.backward()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_(leaves, clip)
        opt.step()
    t0 = time.perf_counter()
    step()                               # warm-up: the first step of a process runs 4 - 10 x slower than the second (first-touch page
    cold = time.perf_counter() - t0      # faults of several GB of activations), which would flatter the GPU
    times = []
    t_start = time.perf_counter()
    while not times or time.perf_counter() - t_start + times[-1] < budget_s:
        t0 = time.perf_counter()
        step()
        times.append(time.perf_counter() - t0)
    times.sort()
    per = times[len(times) // 2]
    return {"value": round(B / per, 4), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", "steps": len(times),
            "s_per_step": round(per, 3), "warmup_step_s": round(cold, 3),
            "sample": f"median of {len(times)} train step(s) of the same workload (batch {B}, {T - 1} lead time, fp32"
                      f"{', clip_grad_norm_' if clip is not None else ''}) after 1 warm-up step, glibc malloc kept from returning "
                      f"memory between steps (mallopt), torch {torch.__version__} CPU, {note}"}


def main_sfno(args):
    line = run_dlwp(args, args.workload, args.steps, args.warmup, init_dist=True)
    if line is not None:
        print(json.dumps(line), flush=True)


def rank_evidence(dist, device, backend, dt, steps):
    """What lets the reader see that N ranks really formed one communicator and ran in step: world size and backend as
    torch.distributed reports them, the RCCL version, one (host, device index, PCI bus id) per rank gathered THROUGH the
    communicator, and every rank's own wall time per step (the line's ms_per_step is their maximum)."""
    import socket
    import torch
    world = dist.get_world_size()
    prop = torch.cuda.get_device_properties(device)
    me = {"rank": dist.get_rank(), "host": socket.gethostname(), "device": device.index, "name": prop.name,
          "pci_bus_id": getattr(prop, "pci_bus_id", None), "ms_per_step": round(dt / steps * 1e3, 4)}
    box = [None] * world
    dist.all_gather_object(box, me)
    ver = None
    try:
        ver = ".".join(str(v) for v in torch.cuda.nccl.version())
    except Exception:          # noqa: BLE001 -- evidence only
        pass
    return {"backend": backend, "rccl_ranks": world if backend == "nccl" else 0, "world_size": world, "rccl_version": ver, "ranks": box}


def run_dlwp(args, workload, steps, warmup, init_dist, roofline=True, cpu=True, batch=None, clip=None):
    """BASELINE configs[2] (sfno) and the supplementary configs[3] / [4] lines (pangu, swin, afno): one step = rollout + MSE +
    backward + all-reduce (N>1) + fused Adam through train_engine.GraphedTrainStep (captured in a hipGraph at N = 1 and for the
    flat reducer; the 28-72 M parameter models use the bucketed reducer launched from backward hooks at N > 1: eager step).
    Returns the JSON line (rank 0) or None."""
    import torch
    import torch.distributed as dist
    from dlwp_benchmark_amd import ddp, dlwpbench, lib as L
    from dlwp_benchmark_amd.train_engine import GraphedTrainStep
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a GPU (the product has no CPU path) [rank {rank} of {world}]")
    device = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(device)
    if world > 1 and init_dist:
        dist.init_process_group("nccl", device_id=device) if args.backend == "nccl" else dist.init_process_group(args.backend)
    w = DLWP_WORKLOADS[workload]
    precision = args.precision or "bf16"
    L.set_gemm_precision(precision)
    # bf16 arithmetic goes with bf16 storage of the GEMM-to-GEMM tensors and of the weight copy the GEMMs read (lib.set_storage);
    # fp32 master weights, gradients, statistics and the residual stream stay fp32.  --storage overrides
    storage = args.storage or (w["storage"] if precision == "bf16" else "fp32")
    L.set_storage(storage)
    B = batch if batch is not None else (args.batch if (args.batch_given and workload == args.workload) else w["batch"])
    lr = w.get("lr", 1e-3)
    if clip is None:
        clip = not args.no_clip
    clip_max_norm = lr if clip else None          # scripts/train.py:133-135: clip_grad_norm_(model.parameters(), current learning rate)
    H, W_, Cg, T = w["H"], w["W"], w["Cg"], w["T"]
    torch.manual_seed(1234)
    model = getattr(dlwpbench, w["cls"])(**w["model"]).to(device).train()
    g = torch.Generator().manual_seed(1234 + rank)
    kw = dict(constants=torch.randn(B, 1, 4, H, W_, generator=g).to(device),
              prescribed=torch.randn(B, T, 1, H, W_, generator=g).to(device),
              prognostic=torch.randn(B, T, Cg, H, W_, generator=g).to(device))
    target = torch.randn(B, T - 1, Cg, H, W_, generator=g).to(device)
    # N > 1, measured on one card (profiles/r04_nograph_lines.jsonl): without the hipGraph the SFNO step (221 launches of ~14 us)
    # drops from 1254 to 739 samples/s, so it keeps the capture and reduces the flat gradient buffer once between the two graph
    # replays; the Pangu / Swin / AFNO steps (kernels of 20 - 300 us) lose <= 1.5 % run eagerly, so they take the bucketed reducer
    # whose all-reduces overlap backward.  --reduce in-graph captures the collective itself (C ABI communicator).
    reduce = args.reduce if world > 1 else ("flat" if getattr(args, "split_graph", False) else "none")
    if reduce == "auto":
        reduce = "flat" if workload == "sfno" else "bucketed"
    bucketed = reduce == "bucketed"
    allreduce, comm = None, None
    if reduce == "flat":
        allreduce = ddp.FlatGradAllReduceChecked()
    elif reduce == "in-graph":
        if args.share_device:
            raise SystemExit("--reduce in-graph needs one GPU per rank (RCCL refuses two ranks on one device)")
        comm = ddp.RcclComm(rank, world)
        allreduce = comm
    step = GraphedTrainStep(model, kw, target, lr=lr, clip_max_norm=clip_max_norm, use_graph=not args.no_graph and not bucketed,
                            allreduce=allreduce, grad_scale=1.0 / world)
    if bucketed:
        step.allreduce = ddp.BucketedGradAllReduce(model, step.grad)
    ddp.broadcast_parameters(step.flat, src=0)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    evidence = None
    if world > 1:
        evidence = rank_evidence(dist, device, args.backend, dt, steps)
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    line = None
    if rank == 0:
        line = {"metric": w["metric"],
                "value": round(world * B * steps / dt, 3), "unit": "samples/s", "n_gpus": world, "steps": steps,
                "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None,
                "dtype": ("bf16" if storage == "bf16" else "bf16 operands, fp32 storage") if precision == "bf16" else "f32",
                "data": "synthetic N(0,1) fields (z-scored WeatherBench shapes), random-init weights (no dataset/checkpoint access)",
                "config": {"workload": w["name"], "per_gpu_batch": B, "global_batch": B * world, "sequence_length": T,
                           "net_calls_per_sample": T - 1, "grid": [H, W_],
                           "n_params": sum(p.numel() for p in model.parameters()),
                           "gemm_operands": precision, "storage": storage, "accumulate": "fp32", "parallelism": f"dp{world}",
                           "learning_rate": lr, "clip_grad_norm": clip_max_norm,
                           "protocol": "src/dlwpbench/configs/training/default.yaml: batch_size 16, clip_gradients true at max_norm = "
                                       "learning rate (scripts/train.py:133-135), Adam" + ("" if clip and B == 16 else
                                       f"; THIS line: per-GPU batch {B}, clipping {'on' if clip else 'off'}"),
                           "hip_graph": not args.no_graph and not bucketed,
                           "grad_reduce": {"none": "none", "flat": "one all-reduce of the flat gradient buffer between the graph replays",
                                           "in-graph": "one all-reduce captured inside the step's graph (dlwp_comm_allreduce)",
                                           "bucketed": "buckets from backward hooks, overlapped with backward (eager step)"}[reduce]},
                "backbone_calls_per_s": round(world * B * steps * (T - 1) / dt, 1), "final_loss": loss.item()}
        if world == 1 and reduce != "none":
            line["config"]["split_graph"] = "N > 1 graph structure at world 1 (no-op reducer between the two graph replays)"
        if bucketed:
            line["config"]["buckets_overlapped"] = getattr(step.allreduce, "overlapped", None)
        if evidence is not None:
            line["ranks"] = evidence
        if world == 1 and not args.no_roofline and roofline:
            if workload == "sfno" and precision == "bf16" and storage == "bf16":
                line["roofline"] = sfno_dominant_probe(device, B)
            elif workload == "sfno":
                line["roofline"] = sfno_gemm_probe(device, B, precision, storage=storage, shape=w["gemm"])
            else:
                line["roofline"] = live_dominant_roofline(step, workload)
        if world == 1 and not args.no_cpu_baseline and cpu:
            del step
            torch.cuda.empty_cache()
            line["cpu_baseline"] = (sfno_cpu_baseline(B, args.cpu_seconds, clip=clip_max_norm) if workload == "sfno" else
                                    dlwp_cpu_baseline(workload, B, args.cpu_seconds, clip=clip_max_norm))
    if comm is not None:
        comm.close()
    if world > 1 and init_dist:
        dist.destroy_process_group()
    return line


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    if args.workload != "fno":
        return main_sfno(args)
    import torch
    import torch.distributed as dist
    from dlwp_benchmark_amd import nsbench

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a GPU (the product has no CPU path) [rank {rank} of {world}]")
    device = torch.device("cuda", 0 if args.share_device else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.backend)

    w = WORKLOAD
    if args.hidden is not None:
        w["hidden_channels"] = args.hidden
        w["name"] = w["name"].replace("(BASELINE configs[1])", f"(BASELINE configs[1] at hidden_channels {args.hidden})")
    if args.T is not None:
        w["T"] = args.T
        w["name"] = w["name"].replace("(BASELINE configs[1]", f"(T={args.T}: {args.T - w['context_size'] + 1} net calls per sample; BASELINE configs[1]")
    B = args.batch
    torch.manual_seed(1234)
    model = nsbench.TFNO2DModule(n_modes=w["n_modes"], in_channels=w["in_channels"],
                                 hidden_channels=w["hidden_channels"], lifting_channels=w["lifting_channels"],
                                 projection_channels=w["projection_channels"], out_channels=w["out_channels"],
                                 n_layers=w["n_layers"], context_size=w["context_size"]).to(device)
    from dlwp_benchmark_amd import ddp
    ddp.broadcast_parameters(model.flat_params.data, src=0)
    opt = model.make_optimizer(lr=1e-3)
    # synthetic trajectories (seeded per rank: every rank trains on its own shard), resident in HBM
    g = torch.Generator().manual_seed(1234 + rank)
    u = torch.randn(B, w["T"] + 1, 1, w["H"], w["W"], generator=g).to(device)
    # the batch is staged in the buffers the captured step reads (where a loader's host-to-device copy would land)
    x, y = model.io_buffers(B, w["T"], w["H"], w["W"], w["teacher_forcing_steps"])
    x.copy_(u[:, :-1])
    y.copy_(u[:, 1:])
    reducer = ddp.FlatGradAllReduce()          # one flat RCCL bucket per step (no-op at world 1)
    allreduce = reducer if world > 1 else None
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
    evidence = None
    if world > 1:
        evidence = rank_evidence(dist, device, args.backend, dt, args.steps)
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
        if evidence is not None:
            line["ranks"] = evidence
            line["config"]["grad_reduce"] = "one all-reduce of the flat gradient buffer between the graph replays"
        if world == 1 and not args.no_roofline:
            line["roofline"] = roofline_probe(device, B)
            if B <= 8 and args.hidden is None:
                # how to read a fraction this low: at the reference's batch size the step is a serial chain of 215 launches of
                # <= 256 workgroups each, so its time does not depend on the batch (profiles/r04_bench_batch_sweep.jsonl) and the
                # dominant kernel moves 4.7 MB per launch; the same kernel at batch 64 is probed beside it
                big = roofline_probe(device, 64, reps=100)
                line["roofline"]["at_batch_64"] = {k: big[k] for k in ("achieved", "peak", "unit", "frac", "bytes_per_launch", "us_per_launch")}
                # ... and the step itself at per-GPU batch 1, timed in this run (same model, a second captured step of that shape):
                # a step time that does not follow the batch is what "launch-latency-bound" means
                x1, y1 = model.io_buffers(1, w["T"], w["H"], w["W"], w["teacher_forcing_steps"])
                x1.copy_(u[:1, :-1])
                y1.copy_(u[:1, 1:])
                for _ in range(5):
                    model.train_step(x1, y1, w["teacher_forcing_steps"], optimizer=opt, use_graph=not args.no_graph)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(40):
                    model.train_step(x1, y1, w["teacher_forcing_steps"], optimizer=opt, use_graph=not args.no_graph)
                torch.cuda.synchronize()
                ms_b1 = (time.perf_counter() - t1) / 40 * 1e3
                ms_b = dt / args.steps * 1e3
                line["roofline"]["regime"] = {
                    "ms_per_step_at_batch_1": round(ms_b1, 4), f"ms_per_step_at_batch_{B}": round(ms_b, 4),
                    "batch_1_over_this": round(ms_b1 / ms_b, 3), "measured": "40 steps at per-GPU batch 1 after 5 warm-ups, same model, in this run",
                    "reading": ("launch-latency-bound: the step is a serial chain of dependent launches of <= 256 workgroups whose time "
                                "barely follows the batch, so the fraction says how much of HBM one such launch can use, not that the "
                                "kernel re-reads data (traffic vs bytes_per_launch)") if ms_b1 > 0.8 * ms_b else
                               "the step time follows the batch: throughput-bound at this batch size"}
            if w["hidden_channels"] <= 64:          # the fused lifting kernel exists for narrow layers only
                line["roofline_mfma"] = mfma_probe(device, B)
            line["roofline_mix"] = mix_probe(device, B)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(B, args.cpu_seconds)
        if world == 1 and not args.no_secondary and args.hidden is None and args.T is None:
            # the other half of BASELINE's metric ("FNO 64^2, SFNO 32x64") in the same driver-timed record: a short run of
            # configs[2] with its own roofline object (python bench.py --workload sfno prints the full line)
            del model, opt
            torch.cuda.empty_cache()
            from dlwp_benchmark_amd import lib as L
            prev_precision, prev_storage = L.load().dlwp_get_gemm_precision(), ("bf16" if L.storage_bf16() else "fp32")
            try:
                # the reference's own dlwpbench protocol: per-GPU batch 16, clip_grad_norm_ at max_norm = learning rate
                sec = run_dlwp(args, "sfno", steps=40, warmup=5, init_dist=False, cpu=True, batch=16, clip=True)
                line["secondary"] = {k: sec[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config",
                                                         "roofline", "cpu_baseline") if k in sec}
                torch.cuda.empty_cache()
                # ... and the round 1 - 4 form of this line (batch 4, no clipping) for continuity
                b4 = run_dlwp(args, "sfno", steps=40, warmup=5, init_dist=False, roofline=False, cpu=False, batch=4, clip=False)
                line["secondary"]["b4_noclip"] = {k: b4[k] for k in ("value", "unit", "steps", "ms_per_step")}
            except Exception as exc:          # noqa: BLE001 -- the headline record above is measured: never lose it to the extra run
                line["secondary"] = {"error": f"{type(exc).__name__}: {exc}"}
            finally:
                L.set_storage("fp32")
                L.set_gemm_precision("bf16" if prev_precision == 1 else "fp32")
                if prev_storage == "bf16" and prev_precision == 1:
                    L.set_storage("bf16")
        if world == 1 and not args.no_tertiary and args.hidden is None and args.T is None:
            # BASELINE configs[3] (Swin, Pangu: 128 x 256, window 7) and configs[4] (FourCastNet AFNO on 721 x 1440) in the same
            # driver-timed record: value, roofline of the kernel that leads each step (live) and a bounded cpu_baseline each
            # (python bench.py --workload swin|pangu|afno721 prints a full line)
            from dlwp_benchmark_amd import lib as L
            line["tertiary"] = []
            left = lambda: args.time_budget - (time.time() - T_PROCESS_START)          # noqa: E731
            for wl in ("swin", "pangu", "afno721"):
                torch.cuda.empty_cache()
                if left() < 60:
                    line["tertiary"].append({"workload": wl, "skipped": f"--time-budget {args.time_budget:.0f} s nearly used up"})
                    continue
                try:
                    ter = run_dlwp(args, wl, steps=20, warmup=3, init_dist=False, cpu=False)
                    line["tertiary"].append({k: ter[k] for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype",
                                                                 "config", "roofline") if k in ter})
                    line["tertiary"][-1]["_wl"] = wl
                except Exception as exc:          # noqa: BLE001 -- never lose the measured record to an extra run
                    line["tertiary"].append({"workload": wl, "error": f"{type(exc).__name__}: {exc}"})
                finally:
                    L.set_storage("fp32")
                    L.set_gemm_precision("fp32")
            # the CPU legs last (a warm-up step + a timed step of each oracle: about a minute altogether), each only while the
            # run's time budget allows: the GPU record above is never put at risk by them
            for rec in line["tertiary"]:
                wl = rec.pop("_wl", None)
                if wl is None or args.no_cpu_baseline:
                    continue
                if left() < CPU_LEG_SECONDS[wl]:
                    rec["cpu_baseline"] = {"skipped": f"{left():.0f} s of --time-budget {args.time_budget:.0f} s left, this leg needs "
                                                      f"~{CPU_LEG_SECONDS[wl]} s; python bench.py --workload {wl} measures it",
                                           "build_container": "profiles/r06_cpu_reference_c4_c5.json"}
                    continue
                try:
                    rec["cpu_baseline"] = dlwp_cpu_baseline(wl, None, 5.0, clip=rec["config"].get("clip_grad_norm"))
                except Exception as exc:          # noqa: BLE001
                    rec["cpu_baseline"] = {"error": f"{type(exc).__name__}: {exc}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
