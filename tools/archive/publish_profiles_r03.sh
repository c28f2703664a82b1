#!/bin/bash
# Copy the summaries of gpurun_out/final_r03/ (tools/collect_profiles_r03.sh a / b) into profiles/ under the round's names.
R=$(cd "$(dirname "$0")/../.." && pwd)
T=r03
F=$R/gpurun_out/final_r03
P=$R/profiles
stats() { ls -t "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cpif() { [ -s "$1" ] && grep -v "amdgpu.ids" "$1" > "$2"; }
cpif "$F/bench_line.json" "$P/${T}_bench_line.json"
cpif "$F/bench_line_T49.json" "$P/${T}_bench_line_T49.json"
cpif "$F/bench_line_h217.json" "$P/${T}_bench_line_hidden217.json"
cpif "$F/bench_batch_sweep.jsonl" "$P/${T}_bench_batch_sweep.jsonl"
for wl in sfno swin pangu afno afno721; do cpif "$F/bench_line_$wl.json" "$P/${T}_bench_line_$wl.json"; done
cpif "$F/models.jsonl" "$P/${T}_models_bench.jsonl"
[ -s "$F/traffic.json" ] && cp "$F/traffic.json" "$P/traffic.json"
cpif "$F/fft_bench.txt" "$P/${T}_fft_bench.txt"
cpif "$F/gemm_bench.txt" "$P/${T}_gemm_bench.txt"
cpif "$F/mfma_valu_coexec.txt" "$P/${T}_mfma_valu_coexec.txt"
s=$(stats prof_bench); [ -n "$s" ] && cp "$s" "$P/${T}_bench_step_kernel_stats.csv"
s=$(stats prof_probe); [ -n "$s" ] && cp "$s" "$P/${T}_spatial_probe_kernel_stats.csv"
for m in sfno afno_fcn pangu_c4 swin_c4; do s=$(stats prof_bf16s_$m); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_${m}_step_kernel_stats.csv"; done
cpif "$F/aten_audit_big.txt" "$P/${T}_aten_audit_big.txt"
cpif "$F/layernorm_probe.txt" "$P/${T}_layernorm_probe_final.txt"
cpif "$F/fft_planar_probe.txt" "$P/${T}_fft_planar_probe.txt"
cpif "$F/bench_line_sfno_b16.json" "$P/${T}_bench_line_sfno_b16.json"
for m in afno pangu; do [ -s "$F/aten_audit_$m.txt" ] && grep -v "Warn\|_warn\|amdgpu.ids" "$F/aten_audit_$m.txt" > "$P/${T}_aten_audit_${m}.txt"; done
# (profiles/r03_published_rmse.json is written from gpurun_out/published_rmse/published_rmse.jsonl with its header note: not overwritten here)
ls -la "$P" | tail -40
