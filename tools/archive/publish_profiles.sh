#!/bin/bash
# Copy the summaries of gpurun_out/final/ (tools/collect_profiles.sh) into profiles/ under the round's names.
#   bash tools/publish_profiles.sh [round tag, default r01]
R=$(cd "$(dirname "$0")/../.." && pwd)
T=${1:-r01}
F=$R/gpurun_out/final
P=$R/profiles
stats() { ls "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cp "$F/bench_line.json" "$P/${T}_bench_line.json"
cp "$F/bench_line_sfno.json" "$P/${T}_bench_line_sfno.json"
cp "$F/bench_line_sfno_fp32.json" "$P/${T}_bench_line_sfno_fp32.json"
cp "$F/models.jsonl" "$P/${T}_models_bench.jsonl"
cp "$F/traffic.json" "$P/traffic.json"
cp "$(stats prof_bench)" "$P/${T}_bench_step_kernel_stats.csv"
cp "$(stats prof_probe)" "$P/${T}_spatial_probe_kernel_stats.csv"
for m in afno swin sfno pangu afno_fcn; do cp "$(stats prof_$m)" "$P/${T}_${m}_step_kernel_stats.csv"; done
ls -la "$P"
