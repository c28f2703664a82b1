set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_token_ops.py tests/test_gpu_swin.py tests/test_gpu_pangu.py tests/test_gpu_afno.py -q -x -m gpu 2>&1 | tail -3
for wl in afno pangu swin; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline; done > gpurun_out/models_new.jsonl 2>&1
python bench.py --workload sfno --steps 40 --warmup 5 --no-roofline --no-cpu-baseline >> gpurun_out/models_new.jsonl 2>&1
