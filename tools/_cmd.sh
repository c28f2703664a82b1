set -e
mkdir -p gpurun_out
for wl in afno pangu; do python bench.py --workload $wl --steps 20 --warmup 3 --no-roofline --no-cpu-baseline; done > gpurun_out/models_new.jsonl 2>&1
