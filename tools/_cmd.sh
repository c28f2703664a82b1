set -e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fft.py tests/test_gpu_afno.py -q -x -m gpu 2>&1 | tail -3
for wl in afno afno721; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline; done > gpurun_out/models_new.jsonl 2>&1
