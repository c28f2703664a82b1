set -e
python tools/bench_models.py afno --steps 20 2>&1 | grep -v amdgpu | cut -c1-200
python bench.py --workload sfno --steps 40 --warmup 5 --no-roofline --no-cpu-baseline | cut -c1-200
python tools/bench_models.py swin --steps 10 2>&1 | grep -v amdgpu | cut -c1-200
