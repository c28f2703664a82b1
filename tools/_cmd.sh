set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pangu.py tests/test_gpu_pad_skip.py -q -x -m gpu 2>&1 | tail -3
python bench.py --workload pangu --steps 20 --warmup 3 --no-roofline --no-cpu-baseline > gpurun_out/models_new.jsonl 2>&1
