python -m pytest tests/test_gpu_sfno.py tests/test_gpu_bf16_storage.py tests/test_gpu_token_ops.py -q -x -m gpu 2>&1 | tail -2
python bench.py --workload sfno --steps 40 --warmup 5 --no-roofline --no-cpu-baseline | cut -c1-200
python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-roofline --no-cpu-baseline | cut -c1-200
