set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bf16_storage.py tests/test_gpu_token_ops.py -q -x -m gpu 2>&1 | tail -3
python tools/bench_gemm.py > gpurun_out/gemm_bench_new.txt 2>&1
for wl in afno pangu swin; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline; done > gpurun_out/models_new.jsonl 2>&1
python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-roofline --no-cpu-baseline >> gpurun_out/models_new.jsonl 2>&1
python bench.py --workload afno721 --steps 10 --warmup 3 --no-roofline --no-cpu-baseline >> gpurun_out/models_new.jsonl 2>&1
