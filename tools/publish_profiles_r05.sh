#!/bin/bash
# Copy the summaries of gpurun_out/final_r05/ (tools/collect_profiles_r05.sh b / a) into profiles/ under the round's names.
R=$(cd "$(dirname "$0")/.." && pwd)
T=r05
F=$R/gpurun_out/final_r05
P=$R/profiles
stats() { ls -t "$F/$1"/*/*kernel_stats.csv 2>/dev/null | head -1; }
cpif() { [ -s "$1" ] && grep -v "amdgpu.ids" "$1" > "$2"; }
cpif "$F/bench_line.json" "$P/${T}_bench_line.json"
cpif "$F/bench_line_T49.json" "$P/${T}_bench_line_T49.json"
cpif "$F/bench_line_h217.json" "$P/${T}_bench_line_hidden217.json"
cpif "$F/bench_batch_sweep.jsonl" "$P/${T}_bench_batch_sweep.jsonl"
for wl in sfno sfno_b4 sfno_split swin pangu afno afno721; do cpif "$F/bench_line_$wl.json" "$P/${T}_bench_line_$wl.json"; done
[ -s "$F/traffic.json" ] && cp "$F/traffic.json" "$P/traffic.json"
cpif "$F/fft_bench.txt" "$P/${T}_fft_bench.txt"
cpif "$F/gemm_epilogue.txt" "$P/${T}_gemm_epilogue.txt"
cpif "$F/sfno_spectral_probe.txt" "$P/${T}_sfno_spectral_probe.txt"
cpif "$F/sfno_stamps.txt" "$P/${T}_sfno_stamps.txt"
cpif "$F/chain_stamps.txt" "$P/${T}_chain_stamps.txt"
cpif "$F/aten_audit_sfno.txt" "$P/${T}_aten_audit_sfno.txt"
for b in b4 b16; do cpif "$F/pmc_sfno_$b/summary.txt" "$P/${T}_sfno_pmc_$b.txt"; done
s=$(stats prof_bench); [ -n "$s" ] && cp "$s" "$P/${T}_bench_step_kernel_stats.csv"
s=$(stats prof_probe); [ -n "$s" ] && cp "$s" "$P/${T}_spatial_probe_kernel_stats.csv"
for m in sfno_b16 sfno afno_fcn pangu_c4 swin_c4; do s=$(stats prof_bf16s_$m); [ -n "$s" ] && cp "$s" "$P/${T}_bf16_storage_${m}_step_kernel_stats.csv"; done
ls -la "$P" | grep "$T" | tail -40
