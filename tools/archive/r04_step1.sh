#!/bin/bash
# round 4, first GPU call: parity of the one-launch SFNO tail + the C3 step with and without it
set -o pipefail
mkdir -p gpurun_out/r04
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_mlp_chain.py tests/test_gpu_sfno.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/step1_tests.txt
cat gpurun_out/r04/step1_tests.txt
python bench.py --workload sfno --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04/sfno_chain.json 2> gpurun_out/r04/sfno_chain.err
DLWP_SFNO_CHAIN=0 python bench.py --workload sfno --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04/sfno_nochain.json 2> gpurun_out/r04/sfno_nochain.err
python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/r04/sfno_chain_b16.json 2> gpurun_out/r04/sfno_chain_b16.err
cat gpurun_out/r04/sfno_chain.json gpurun_out/r04/sfno_nochain.json gpurun_out/r04/sfno_chain_b16.json | cut -c1-400
rocprofv3 --kernel-trace --stats -d gpurun_out/r04/prof_sfno -o sfno -- python3 bench.py --workload sfno --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/r04/prof_sfno.log 2>&1
python tools/prof_summary.py gpurun_out/r04/prof_sfno 2>/dev/null | head -40
