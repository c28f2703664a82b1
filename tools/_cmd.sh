set -e
python tools/bench_gemm.py 2>&1 | grep -E "C5.*(y=|gx=)" | cut -c1-150
echo nostore
DLWP_GEMM_EXP_NOSTORE=1 python tools/bench_gemm.py 2>&1 | grep -E "C5.*(y=|gx=)" | cut -c1-150
