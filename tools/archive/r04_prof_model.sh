#!/bin/bash
# rocprofv3 kernel stats of one bench_models.py model (bf16 operands + storage) -> gpurun_out/r04/<tag>_kernel_stats.csv
# usage: r04_prof_model.sh <model> <tag> [ENV=VALUE ...]
set -o pipefail
model=$1; tag=$2; shift 2
for kv in "$@"; do export "$kv"; done
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r04/prof_$tag
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04/prof_$tag -- python3 tools/bench_models.py $model --steps 3 --precision bf16 --storage bf16 > gpurun_out/r04/prof_$tag.log 2>&1
find gpurun_out/r04/prof_$tag -name "*_kernel_trace.csv" -delete
cp $(find gpurun_out/r04/prof_$tag -name "*_kernel_stats.csv" | head -1) gpurun_out/r04/${tag}_kernel_stats.csv
python tools/prof_summary.py gpurun_out/r04/prof_$tag 12
