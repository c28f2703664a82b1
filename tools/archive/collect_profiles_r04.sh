#!/bin/bash
# Round-4 evidence run on the GPU box, in two gpurun calls (each under the 1200 s limit); everything lands in
# gpurun_out/final_r04/ and tools/publish_profiles_r04.sh copies the summaries into profiles/.
#   gpurun --timeout 1190 -- 'bash tools/collect_profiles_r04.sh a'     headline: bench lines, step profile, PMC traffic, B sweep
#   gpurun --timeout 1190 -- 'bash tools/collect_profiles_r04.sh b'     C3-C5 lines and step profiles, FFT / GEMM / attention probes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/final_r04
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() { d=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- python3 "$@" > /dev/null 2>&1; find $O/$d -name "*_kernel_trace.csv" -delete; }
if [ "$1" = "a" ]; then
  timeout 600 python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
  echo "bench done"; cut -c1-300 $O/bench_line.json
  timeout 200 python3 $R/bench.py --T 49 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > $O/bench_line_T49.json 2>> $O/bench.err
  timeout 300 python3 $R/bench.py --hidden 217 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_line_h217.json 2>> $O/bench.err
  for b in 1 2 8 16 32 64; do timeout 200 python3 $R/bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary >> $O/bench_batch_sweep.jsonl 2>> $O/bench.err; done
  prof prof_bench $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary
  prof prof_probe $R/tools/probe_spatial.py
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write fno_spatial_kernel $O/traffic.json "fno_spatial_kernel<2,1>" > $O/traffic.log 2>&1
  find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*_kernel_trace.csv" -delete
  timeout 60 $R/tools/micro/mfma_valu_coexec > $O/mfma_valu_coexec.txt 2>&1
  timeout 400 python3 $R/tools/bench_fft.py > $O/fft_bench.txt 2>&1
  cat $O/traffic.log; cut -c1-200 $O/bench_line_T49.json
else
  timeout 400 python3 $R/bench.py --workload sfno > $O/bench_line_sfno.json 2> $O/bench_b.err
  for wl in swin pangu afno afno721; do timeout 300 python3 $R/bench.py --workload $wl --steps 20 --warmup 3 > $O/bench_line_$wl.json 2>> $O/bench_b.err; echo "$wl done"; done
  prof prof_bf16s_sfno $R/bench.py --workload sfno --steps 20 --warmup 3 --no-cpu-baseline --no-roofline
  for m in afno_fcn pangu_c4 swin_c4; do prof prof_bf16s_$m $R/tools/bench_models.py $m --steps 3 --precision bf16 --storage bf16; done
  timeout 300 python3 $R/tools/bench_gemm.py > $O/gemm_bench.txt 2>&1
  timeout 100 python3 $R/tools/bench_gemm_epilogue.py 2>&1 | grep "T=" > $O/gemm_epilogue.txt
  timeout 600 python3 $R/tools/bench_models.py all --steps 10 > $O/models.jsonl 2> $O/models.err
  for m in afno pangu; do timeout 200 python3 $R/tools/aten_audit.py $m > $O/aten_audit_$m.txt 2>&1; done
  for m in afno_fcn pangu_c4 swin_c4 sfno; do timeout 200 python3 $R/tools/aten_audit.py $m --precision bf16 --storage bf16 2>&1 | grep "^==\|^ " ; done > $O/aten_audit_big.txt
  timeout 200 python3 $R/tools/probe_layernorm.py > $O/layernorm_probe.txt 2>&1
  timeout 200 python3 $R/tools/probe_fft_ib.py > $O/fft_planar_probe.txt 2>&1
  (for v in 1 0; do DLWP_WINATTN_BWD2PASS=$v timeout 100 python3 $R/tools/probe_winattn_c4.py 2>&1 | grep "B_=" | sed "s/$/ BWD2PASS=$v/"; done; for wg in 456 798; do PROBE_FIRST=1 DLWP_WINATTN_WG_BWD=$wg timeout 100 python3 $R/tools/probe_winattn_c4.py 2>&1 | grep "B_="; done; for dbg in 1 3 8 32; do PROBE_FIRST=1 DLWP_WINATTN_DBG=$dbg timeout 100 python3 $R/tools/probe_winattn_c4.py 2>&1 | grep "B_="; done) > $O/winattn_probe.txt
  if [ -f $R/dlwp_benchmark_amd/libdlwpmi_stamps.so ]; then
    timeout 100 python3 $R/tools/probe_stamps_winattn1p.py 2>&1 | grep -v amdgpu.ids > $O/winattn_1p_stamps.txt
    timeout 100 python3 $R/tools/probe_stamps_chain.py 2>&1 | grep -v amdgpu.ids > $O/chain_stamps.txt
  fi
  timeout 200 python3 $R/tools/aten_audit.py sfno --precision bf16 --storage bf16 2>&1 | grep "^==\|^ " > $O/aten_audit_sfno.txt
  timeout 100 python3 $R/bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-roofline --no-cpu-baseline > $O/bench_line_sfno_b16.json 2>> $O/bench_b.err
  for f in sfno swin pangu afno afno721; do cut -c1-220 $O/bench_line_$f.json; done
fi
echo "=== done $1"
