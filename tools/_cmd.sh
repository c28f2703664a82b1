set -e
python -m pytest tests/test_gpu_bf16_storage.py -q -x -m gpu 2>&1 | tail -2
for wl in afno pangu swin; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline; done > gpurun_out/models_new.jsonl 2>&1
python bench.py --workload sfno --steps 40 --warmup 5 --no-roofline --no-cpu-baseline >> gpurun_out/models_new.jsonl 2>&1
python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-roofline --no-cpu-baseline >> gpurun_out/models_new.jsonl 2>&1
