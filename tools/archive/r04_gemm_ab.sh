#!/bin/bash
# round 4: A/B of the GEMM epilogue changes (packed GELU in epilogue_group; 256-tile kernel issues the next tile's prologue before its epilogue)
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
for v in old new; do
  f=libdlwpmi.so; [ $v = old ] && f=libdlwpmi_old.so
  [ -f dlwp_benchmark_amd/$f ] || continue
  echo "== $v" | tee -a $O/gemm_epilogue_ab.txt
  DLWP_LIB_FILE=$f timeout -k 10 120 python tools/bench_gemm_epilogue.py 2>&1 | grep "T=" | tee -a $O/gemm_epilogue_ab.txt || exit 1
  echo "== $v, 256-tile kernel forced (DLWP_GEMM_P8=1)" | tee -a $O/gemm_epilogue_ab.txt
  DLWP_GEMM_P8=1 DLWP_LIB_FILE=$f timeout -k 10 120 python tools/bench_gemm_epilogue.py 2>&1 | grep "T=" | tee -a $O/gemm_epilogue_ab.txt || exit 1
done
