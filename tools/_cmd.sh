python -m pytest tests/test_gpu_token_ops.py -q -x -m gpu 2>&1 | tail -2
python tools/probe_layernorm.py 2>&1 | grep -E "fwd"
for wl in afno pangu; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline | cut -c1-200; done
