#!/bin/bash
# Copy the summaries of gpurun_out/final_r06/ (tools/collect_profiles_r06.sh b / a) into profiles/ under the round's names.
R=$(cd "$(dirname "$0")/.." && pwd)
T=r06
F=$R/gpurun_out/final_r06
P=$R/profiles
stats() { ls -t "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cpif() { [ -s "$1" ] && grep -v "amdgpu.ids" "$1" > "$2"; }
cpif "$F/bench_line.json" "$P/${T}_bench_line.json"
cpif "$F/bench_line_T49.json" "$P/${T}_bench_line_T49.json"
cpif "$F/bench_line_h217.json" "$P/${T}_bench_line_hidden217.json"
cpif "$F/bench_batch_sweep.jsonl" "$P/${T}_bench_batch_sweep.jsonl"
for wl in sfno sfno_b4 swin pangu afno afno721; do cpif "$F/bench_line_$wl.json" "$P/${T}_bench_line_$wl.json"; done
[ -s "$F/traffic.json" ] && cp "$F/traffic.json" "$P/traffic.json"
s=$(stats prof_bench); [ -n "$s" ] && cp "$s" "$P/${T}_bench_step_kernel_stats.csv"
s=$(stats prof_probe); [ -n "$s" ] && cp "$s" "$P/${T}_spatial_probe_kernel_stats.csv"
for m in sfno_b16 sfno; do s=$(stats prof_bf16s_$m); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_${m}_step_kernel_stats.csv"; done
s=$(stats prof_bf16s_swin); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_swin_c4_step_kernel_stats.csv"
s=$(stats prof_bf16s_pangu); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_pangu_c4_step_kernel_stats.csv"
s=$(stats prof_bf16s_afno); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_afno_fcn_step_kernel_stats.csv"
s=$(stats prof_bf16s_afno721); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_afno721_step_kernel_stats.csv"
ls -la "$P" | grep "$T" | tail -40
