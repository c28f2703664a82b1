#!/bin/bash
# fp32 / bf16-operand / bf16-storage train-step throughput of the C3-C5 scale models (supplementary; one gpurun call)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/bf16_r02.jsonl
: > $O
for m in sfno swin_c4 pangu_c4 afno_fcn; do
  timeout -k 10 280 python3 $R/tools/bench_models.py $m --steps 5 >> $O 2>/dev/null
  timeout -k 10 280 python3 $R/tools/bench_models.py $m --steps 5 --precision bf16 >> $O 2>/dev/null
  timeout -k 10 280 python3 $R/tools/bench_models.py $m --steps 5 --precision bf16 --storage bf16 >> $O 2>/dev/null
  echo "$m done"
done
cat $O
