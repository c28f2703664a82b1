#!/bin/bash
# Round-6 evidence run on the GPU box, in two gpurun calls (each under the 1200 s limit); everything lands in gpurun_out/final_r06/ and
# tools/publish_profiles_r06.sh copies the summaries into profiles/.
#   gpurun --timeout 1190 -- 'bash tools/collect_profiles_r06.sh b'   step profiles (rocprofv3 --kernel-trace --stats) of every workload of the
#                                                                     bench line: bench.py picks each roofline kernel from these tables
#   (publish)
#   gpurun --timeout 1190 -- 'bash tools/collect_profiles_r06.sh a'   the bench line itself, per-workload lines, PMC traffic of the headline kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/final_r06
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
prof() { d=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -- python3 "$@" > /dev/null 2>&1; find $O/$d -name "*_kernel_trace.csv" -delete; }
if [ "$1" = "b" ]; then
  prof prof_bench $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-tertiary
  prof prof_bf16s_sfno_b16 $R/bench.py --workload sfno --steps 20 --warmup 3 --no-cpu-baseline --no-roofline
  prof prof_bf16s_sfno $R/bench.py --workload sfno --batch 4 --no-clip --steps 20 --warmup 3 --no-cpu-baseline --no-roofline
  for wl in swin pangu afno afno721; do prof prof_bf16s_$wl $R/bench.py --workload $wl --steps 10 --warmup 3 --no-cpu-baseline --no-roofline; echo "$wl profiled"; done
  prof prof_probe $R/tools/probe_spatial.py
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
  python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write fno_spatial_kernel $O/traffic.json "fno_spatial_kernel<2,1>" > $O/traffic.log 2>&1
  find $O -name "*counter_collection.csv" -size +4M -delete; find $O -name "*_kernel_trace.csv" -delete
  cat $O/traffic.log
  for d in prof_bench prof_bf16s_sfno_b16 prof_bf16s_swin prof_bf16s_pangu prof_bf16s_afno prof_bf16s_afno721; do f=$(ls -t $O/$d/*/*kernel_stats.csv 2>/dev/null | head -1); echo "== $d"; [ -n "$f" ] && python3 $R/tools/kernel_table.py $f 1 4; done
else
  ( time timeout 560 python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench.err ) 2> $O/bench.time
  echo "bench done"; cat $O/bench.time; cut -c1-300 $O/bench_line.json
  timeout 200 python3 $R/bench.py --T 49 --steps 50 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-tertiary > $O/bench_line_T49.json 2>> $O/bench.err
  timeout 300 python3 $R/bench.py --hidden 217 --steps 30 --warmup 5 --no-cpu-baseline --no-secondary --no-tertiary > $O/bench_line_h217.json 2>> $O/bench.err
  for b in 1 2 8 16 32 64; do timeout 200 python3 $R/bench.py --batch $b --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-secondary --no-tertiary >> $O/bench_batch_sweep.jsonl 2>> $O/bench.err; done
  timeout 400 python3 $R/bench.py --workload sfno > $O/bench_line_sfno.json 2>> $O/bench.err
  timeout 200 python3 $R/bench.py --workload sfno --batch 4 --no-clip > $O/bench_line_sfno_b4.json 2>> $O/bench.err
  for wl in swin pangu afno afno721; do timeout 300 python3 $R/bench.py --workload $wl --steps 20 --warmup 3 --cpu-seconds 5 > $O/bench_line_$wl.json 2>> $O/bench.err; echo "$wl done"; done
  for f in sfno sfno_b4 swin pangu afno afno721; do cut -c1-200 $O/bench_line_$f.json; done
fi
echo "=== done $1"
