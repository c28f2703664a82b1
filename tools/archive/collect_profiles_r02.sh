#!/bin/bash
# Round-2 evidence run on the GPU box (one gpurun call): bench lines (headline, hidden 217, SFNO), rocprofv3 kernel statistics of
# the headline step / the wide step / the probes / the FFT kernels, PMC traffic passes (FETCH_SIZE and WRITE_SIZE in SEPARATE
# passes, MI355X_MICROARCH.md), supplementary model benches.  Everything lands in gpurun_out/final_r02/.
#   gpurun --timeout 1190 -- 'bash tools/collect_profiles_r02.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/final_r02
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
echo "bench done"; tail -c 400 $O/bench_line.json
timeout 300 python3 $R/bench.py --hidden 217 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_line_h217.json 2>> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h217 -- python3 $R/bench.py --hidden 217 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > /dev/null 2>&1
echo "step profiles done"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mix -- python3 $R/tools/probe_mix.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mix217 -- python3 $R/tools/probe_mix.py 217 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write fno_spatial_kernel $O/traffic.json "fno_spatial_kernel<2,1>" > $O/traffic.log 2>&1
echo "probes done"
timeout 400 python3 $R/tools/bench_fft.py > $O/fft_bench.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fft -- python3 $R/tools/probe_fft.py > /dev/null 2>&1
timeout 400 python3 $R/bench.py --workload sfno > $O/bench_line_sfno.json 2> $O/bench_sfno.err
for wl in swin pangu afno; do timeout 300 python3 $R/bench.py --workload $wl --steps 20 --warmup 3 > $O/bench_line_$wl.json 2>> $O/bench_sfno.err; done
timeout 900 python3 $R/tools/bench_models.py all --steps 10 > $O/models.jsonl 2> $O/models.err
for m in fno_dlwp tfno_dlwp fno_ctx swin_dlwp; do timeout 250 python3 $R/tools/bench_models.py $m --steps 20 >> $O/models.jsonl 2>> $O/models.err; done
timeout 400 python3 $R/tools/bench_models.py afno_c5p1 --steps 3 >> $O/models.jsonl 2>> $O/models.err
timeout 400 python3 $R/tools/bench_models.py afno_c5p1 --steps 3 --precision bf16 --storage bf16 >> $O/models.jsonl 2>> $O/models.err
for m in afno swin pangu; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python3 $R/tools/bench_models.py $m --steps 5 > /dev/null 2>&1
done
# round-2 additions: ATen audit of the three C1-size steps, GEMM / window-attention micro-benchmarks, fp32 / bf16-operand /
# bf16-storage lines of the C3-C5 scale models and the kernel statistics of the bf16-storage C4 / C5 steps
for m in afno swin pangu; do timeout 200 python3 $R/tools/aten_audit.py $m > $O/aten_audit_$m.txt 2>&1; done
timeout 300 python3 $R/tools/bench_gemm.py > $O/gemm_bench.txt 2>&1
DLWP_GEMM_TILE=128 timeout 300 python3 $R/tools/bench_gemm.py > $O/gemm_bench_tile128.txt 2>&1
timeout 300 python3 $R/tools/probe_winattn.py > $O/winattn_probe.txt 2>&1
timeout 900 bash $R/tools/bench_bf16_r02.sh > /dev/null 2>&1; cp $R/gpurun_out/bf16_r02.jsonl $O/bf16_models.jsonl
for m in afno_fcn pangu_c4 swin_c4; do
  timeout 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bf16s_$m -- python3 $R/tools/bench_models.py $m --steps 3 --precision bf16 --storage bf16 > /dev/null 2>&1
done
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +4M -delete
echo "=== lines"; cat $O/bench_line.json; cat $O/bench_line_h217.json; cat $O/traffic.log; cat $O/models.jsonl
