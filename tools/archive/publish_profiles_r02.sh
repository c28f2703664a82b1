#!/bin/bash
# Copy the summaries of gpurun_out/final_r02/ (tools/collect_profiles_r02.sh) into profiles/ under the round's names.
R=$(cd "$(dirname "$0")/../.." && pwd)
T=r02
F=$R/gpurun_out/final_r02
P=$R/profiles
stats() { ls -t "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cp "$F/bench_line.json" "$P/${T}_bench_line.json"
cp "$F/bench_line_h217.json" "$P/${T}_bench_line_hidden217.json"
cp "$F/bench_line_sfno.json" "$P/${T}_bench_line_sfno.json"
for wl in swin pangu afno; do [ -s "$F/bench_line_$wl.json" ] && cp "$F/bench_line_$wl.json" "$P/${T}_bench_line_$wl.json"; done
cp "$F/models.jsonl" "$P/${T}_models_bench.jsonl"
cp "$F/traffic.json" "$P/traffic.json"
cp "$F/fft_bench.txt" "$P/${T}_fft_bench.txt"
cp "$(stats prof_bench)" "$P/${T}_bench_step_kernel_stats.csv"
cp "$(stats prof_h217)" "$P/${T}_hidden217_step_kernel_stats.csv"
cp "$(stats prof_probe)" "$P/${T}_spatial_probe_kernel_stats.csv"
cp "$(stats prof_mix)" "$P/${T}_mix_probe_kernel_stats.csv"
cp "$(stats prof_mix217)" "$P/${T}_mix_probe_hidden217_kernel_stats.csv"
cp "$(stats prof_fft)" "$P/${T}_fft_kernel_stats.csv"
for m in afno swin pangu; do
  cp "$(stats prof_$m)" "$P/${T}_${m}_step_kernel_stats.csv"
  grep -v "Warn\|_warn\|amdgpu.ids" "$F/aten_audit_$m.txt" > "$P/${T}_aten_audit_${m}.txt"
done
grep -v amdgpu.ids "$F/gemm_bench.txt" > "$P/${T}_gemm_bench.txt"
grep -v amdgpu.ids "$F/gemm_bench_tile128.txt" | sed 's/^/[DLWP_GEMM_TILE=128] /' >> "$P/${T}_gemm_bench.txt"
grep -v amdgpu.ids "$F/winattn_probe.txt" > "$P/${T}_winattn_probe.txt"
cp "$F/bf16_models.jsonl" "$P/${T}_bf16_models_bench.jsonl"
for m in afno_fcn pangu_c4 swin_c4; do cp "$(stats prof_bf16s_$m)" "$P/${T}_bf16_storage_${m}_step_kernel_stats.csv"; done
ls -la "$P"
