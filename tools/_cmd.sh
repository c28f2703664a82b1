python bench.py --workload swin --steps 20 --warmup 3 --no-roofline --no-cpu-baseline | cut -c1-200
DLWP_GEMM_TILE=128 python bench.py --workload swin --steps 20 --warmup 3 --no-roofline --no-cpu-baseline | cut -c1-200
