python -m pytest tests/test_gpu_afno.py tests/test_gpu_token_ops.py tests/test_gpu_bf16_storage.py -q -x -m gpu 2>&1 | tail -2
for wl in afno afno721; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline | cut -c1-200; done
