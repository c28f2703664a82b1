#!/bin/bash
# per-kernel statistics of the headline step on the GPU box: tools/kstats.sh <tag> [bench args]; prints the top rows of the
# rocprofv3 kernel_stats CSV (name, calls, total, average ns, percentage) and leaves the CSV under gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=$1; shift
O=$R/gpurun_out/$tag
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps ${KSTEPS:-50} --warmup 5 --no-cpu-baseline --no-roofline "$@" > $O/line.json 2> $O/err.txt
find $O -name "*_kernel_trace.csv" -delete
f=$(find $O -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f'{n[:70]:70s} {int(r["Calls"]):6d} {float(r["AverageNs"]) / 1e3:8.2f} us {float(r["Percentage"]):6.2f} %')
print(f"total {tot / 1e6:.2f} ms over 55 steps = {tot / 55e6:.4f} ms/step")
PY
cat $O/line.json | cut -c1-200
