#!/bin/bash
# Copy the summaries of gpurun_out/final_r02/ (tools/collect_profiles_r02.sh) into profiles/ under the round's names.
R=$(cd "$(dirname "$0")/.." && pwd)
T=r02
F=$R/gpurun_out/final_r02
P=$R/profiles
stats() { ls "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cp "$F/bench_line.json" "$P/${T}_bench_line.json"
cp "$F/bench_line_h217.json" "$P/${T}_bench_line_hidden217.json"
cp "$F/bench_line_sfno.json" "$P/${T}_bench_line_sfno.json"
cp "$F/models.jsonl" "$P/${T}_models_bench.jsonl"
cp "$F/traffic.json" "$P/traffic.json"
cp "$F/fft_bench.txt" "$P/${T}_fft_bench.txt"
cp "$(stats prof_bench)" "$P/${T}_bench_step_kernel_stats.csv"
cp "$(stats prof_h217)" "$P/${T}_hidden217_step_kernel_stats.csv"
cp "$(stats prof_probe)" "$P/${T}_spatial_probe_kernel_stats.csv"
cp "$(stats prof_mix)" "$P/${T}_mix_probe_kernel_stats.csv"
cp "$(stats prof_mix217)" "$P/${T}_mix_probe_hidden217_kernel_stats.csv"
cp "$(stats prof_fft)" "$P/${T}_fft_kernel_stats.csv"
for m in afno swin pangu; do cp "$(stats prof_$m)" "$P/${T}_${m}_step_kernel_stats.csv"; done
ls -la "$P"
