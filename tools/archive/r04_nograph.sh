#!/bin/bash
# round 4: what N > 1 runs on one card -- graphed vs eager (--no-graph) lines of the C3 - C5 workloads, and the N = 2 share-device runs
set -o pipefail
O=gpurun_out/r04/n2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in sfno pangu swin afno; do
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-roofline > $O/graph_$w.json 2> $O/graph_$w.err
  python bench.py --workload $w --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --no-graph > $O/nograph_$w.json 2> $O/nograph_$w.err
  echo "$w graph: $(python -c "import json;d=json.load(open('$O/graph_$w.json'));print(d['value'], d['ms_per_step'])")  no-graph: $(python -c "import json;d=json.load(open('$O/nograph_$w.json'));print(d['value'], d['ms_per_step'])")"
done
# two ranks on one card (gloo): graphed step + flat reduce (default) and eager + bucketed overlap, Pangu and SFNO
for w in pangu sfno; do
  for r in flat bucketed; do
    MASTER_ADDR=127.0.0.1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --share-device --workload $w --reduce $r --steps 10 --warmup 2 --no-cpu-baseline --no-roofline > $O/n2_${w}_$r.json 2> $O/n2_${w}_$r.err
    echo "$w N=2 $r: $(grep '^{' $O/n2_${w}_$r.json | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'], d['ms_per_step'], [r['ms_per_step'] for r in d['ranks']['ranks']])")"
  done
done
