DLWP_BENCH_GEMM_ONLY=gW python tools/bench_gemm_graph.py 2>&1 | grep gW | tr '\n' ';'; echo
python -m pytest tests/test_gpu_bf16_storage.py -q -x -m gpu 2>&1 | tail -2
for wl in pangu afno; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline | cut -c1-200; done
