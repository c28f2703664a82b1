#!/bin/bash
# PMC passes (fabric bytes, L2 hit rate) over the stand-alone spectral kernels of the SFNO block: tools/pmc_sfno.sh <tag> [B]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/$1; B=${2:-4}
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=$O/$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -- python3 $R/tools/probe_stamps_sfno.py --nostamps $B > $O/log.txt 2>&1
  find $d -name "*_kernel_trace.csv" -delete
done
python3 $R/tools/pmc_kernels.py $O sht_ dhconv_apply | tee $O/summary.txt
find $O -name "*counter_collection.csv" -size +2M -delete
