#!/bin/bash
# rocprofv3 kernel stats of the C3 (SFNO) bench step -> gpurun_out/r04/<tag>_kernel_stats.csv
set -o pipefail
tag=${1:-sfno}; shift
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_$tag -- python3 bench.py --workload sfno --steps 20 --warmup 5 --no-cpu-baseline --no-roofline "$@" > gpurun_out/r04/prof_$tag.log 2>&1
find gpurun_out/r04/prof_$tag -name "*_kernel_trace.csv" -delete
cp $(find gpurun_out/r04/prof_$tag -name "*_kernel_stats.csv" | head -1) gpurun_out/r04/${tag}_kernel_stats.csv
python tools/prof_summary.py gpurun_out/r04/prof_$tag 40
