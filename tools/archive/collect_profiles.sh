#!/bin/bash
# Round-end evidence run on the GPU box (one gpurun call): bench line, rocprofv3 kernel statistics of the headline step and
# of the roofline probe, PMC traffic passes, and the supplementary model benches.  Everything lands in gpurun_out/final/.
#   gpurun --timeout 1500 -- 'bash tools/collect_profiles.sh'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/final
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $O/bench_line.json 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_probe -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/probe_spatial.py > /dev/null 2>&1
python3 $R/tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write fno_spatial_kernel $O/traffic.json "fno_spatial_kernel<2,1>" > $O/traffic.log 2>&1
timeout 400 python3 $R/bench.py --workload sfno > $O/bench_line_sfno.json 2> $O/bench_sfno.err
timeout 300 python3 $R/bench.py --workload sfno --precision fp32 --no-cpu-baseline > $O/bench_line_sfno_fp32.json 2>> $O/bench_sfno.err
timeout 600 python3 $R/tools/bench_models.py all --steps 10 > $O/models.jsonl 2> $O/models.err
for m in swin_dlwp swin_c4 pangu_c4 afno_fcn; do
  timeout 300 python3 $R/tools/bench_models.py $m --steps 5 2>> $O/models.err | grep '"model"' >> $O/models.jsonl
done
for m in sfno afno_fcn swin_c4 pangu_c4; do
  timeout 300 python3 $R/tools/bench_models.py $m --steps 5 --precision bf16 2>> $O/models.err | grep '"model"' >> $O/models.jsonl
done
for m in afno swin sfno pangu afno_fcn; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$m -- python3 $R/tools/bench_models.py $m --steps 5 > /dev/null 2>&1
done
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -size +4M -delete
cat $O/bench_line.json; cat $O/bench_line_sfno.json; cat $O/traffic.log; cat $O/models.jsonl
