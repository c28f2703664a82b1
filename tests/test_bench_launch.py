"""bench.py host logic that needs no GPU: the N > 1 launch guard (round-5 verdict: `python bench.py --gpus 8` without a launcher
silently benchmarked ONE GPU and printed "n_gpus": 1) and the helpers that pick the roofline kernel from a committed rocprofv3
step table (profiles/*_step_kernel_stats.csv)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["HIP_VISIBLE_DEVICES"] = ""          # the ranks of this test must not find a GPU wherever it runs
    env["CUDA_VISIBLE_DEVICES"] = ""
    return env


def test_gpus_n_without_launcher_and_no_spawn_fails_with_the_launcher_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-spawn"], env=_env(),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node=2" in r.stderr and "WORLD_SIZE" in r.stderr
    assert '"n_gpus"' not in r.stdout          # no record at all, certainly not a world-1 one


def test_gpus_n_without_launcher_starts_n_ranks():
    """The parent (which never imports torch) starts torch.distributed.run with two ranks; without a GPU each rank stops at
    bench.py's own "needs a GPU" exit, which names its rank and the world size it was given: the ranks were real."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo", "--no-cpu-baseline",
                        "--no-roofline"], env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                    # no GPU here: the ranks cannot run the product
    assert "starting" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert "of 2]" in r.stderr and "needs a GPU" in r.stderr
    assert '"n_gpus": 1' not in r.stdout


def test_world_size_mismatch_is_refused():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_kernel_short_name_and_committed_tables():
    sys.path.insert(0, ROOT)
    import bench
    f = bench.kernel_short_name
    assert f("void (anonymous namespace)::gemm_glds_kernel<false, 32>((anonymous namespace)::GemmDev)") == "gemm_glds_kernel<false, 32>"
    assert f("(anonymous namespace)::wgrad_multi_kernel((anonymous namespace)::WgDev)") == "wgrad_multi_kernel"
    assert f("void (anonymous namespace)::winattn_lds_bwd1p_kernel<2, 8, true>((anonymous namespace)::WsDev)") == \
        "winattn_lds_bwd1p_kernel<2, 8, true>"
    assert f("void (anonymous namespace)::layernorm_bwd_vec_kernel<32>(float const*, float const*, int)") == "layernorm_bwd_vec_kernel<32>"
    # every workload of the tertiary list resolves to a committed table whose leader is a kernel the library's accounting names
    for wl in ("swin", "pangu"):
        path, rows = bench.committed_step_table(wl)
        assert path is not None and rows and rows[0][1] > 5.0, wl
        assert "(" not in rows[0][0] and "anonymous" not in rows[0][0]
