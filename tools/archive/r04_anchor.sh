#!/bin/bash
# round 4: seeds of the external anchor (tools/published_rmse.py), several trainings side by side on the one card.
# usage: r04_anchor.sh <stop-epoch> "<hidden>:<seed>:<alpha> ..."
set -o pipefail
STOP=$1; shift
O=gpurun_out/published_rmse_r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
pids=()
for spec in $1; do
  IFS=: read h s a si <<< "$spec"; si=${si:-1.0}
  python tools/published_rmse.py --phase train --hidden $h --seed $s --alpha $a --spectral-init-scale $si --stop-epoch $STOP --out $O > $O/train_h${h}_s${s}_a${a}_i${si}_to$STOP.log 2>&1 &
  pids+=($!)
done
# progress lines keep the call alive
while true; do
  alive=0
  for p in "${pids[@]}"; do kill -0 $p 2>/dev/null && alive=1; done
  [ $alive = 0 ] && break
  sleep 60; echo "anchor runs alive at $(date +%T)"; tail -qn1 $O/train_*_to$STOP.log | cut -c1-150
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
if [ "$STOP" = "500" ]; then
  for spec in $1; do
    IFS=: read h s a si <<< "$spec"; si=${si:-1.0}
    python tools/published_rmse.py --phase eval --hidden $h --seed $s --alpha $a --spectral-init-scale $si --out $O > $O/eval_h${h}_s${s}_a${a}_i${si}.log 2>&1 || rc=1
    tail -n2 $O/eval_h${h}_s${s}_a${a}_i${si}.log | cut -c1-400
  done
fi
exit $rc
