#!/bin/bash
# round 4: registry / slab / packed-GELU checks, headline and SFNO benches, non-temporal weight-load variant
set -o pipefail
O=gpurun_out/r04; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fno.py tests/test_gpu_sfno.py tests/test_gpu_sfno_io.py tests/test_gpu_mlp_chain.py tests/test_gpu_bf16_storage.py tests/test_gpu_swin.py tests/test_gpu_token_ops.py -x -q -m gpu 2>&1 | tail -4 || exit 1
python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-secondary > $O/headline_packed.json 2> $O/headline_packed.err
python -c "import json;d=json.load(open('$O/headline_packed.json'));print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma']['frac'])"
python bench.py --workload sfno --steps 40 --warmup 5 --no-cpu-baseline > $O/sfno_f.json 2> $O/sfno_f.err
python -c "import json;d=json.load(open('$O/sfno_f.json'));print('sfno', d['value'], d['ms_per_step'], d['roofline']['us_per_launch'], d['roofline']['hbm']['frac'], d['roofline']['mfma']['frac'])"
DLWP_LIB_FILE=libdlwpmi_nt.so python bench.py --workload sfno --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/sfno_f_nt.json 2> $O/sfno_f_nt.err
python -c "import json;d=json.load(open('$O/sfno_f_nt.json'));print('sfno nt', d['value'], d['ms_per_step'])"
python bench.py --workload sfno --batch 16 --steps 40 --warmup 5 --no-cpu-baseline --no-roofline > $O/sfno_f_b16.json 2> $O/sfno_f_b16.err
python -c "import json;d=json.load(open('$O/sfno_f_b16.json'));print('sfno b16', d['value'], d['ms_per_step'])"
