for f in 0 1; do echo "== DLWP_SHT_FUSED=$f"; DLWP_SHT_FUSED=$f python bench.py --workload sfno --steps 40 --warmup 5 --no-roofline --no-cpu-baseline | cut -c1-200
DLWP_SHT_FUSED=$f python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-roofline --no-cpu-baseline | cut -c1-200; done
