#!/bin/bash
# round 4, call 5y: segment-reducer backward with four rows in flight per lane group: operator / net / config tests, C5 and headline bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5y
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_configs_gpu.py tests/test_compgcn_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python bench.py --workload c5_fixed_cell --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 1; }
python bench.py --no-cpu-baseline --no-c5 --no-caller-leg --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5y/bench_c5.json')); print('c5', d['ms_per_step'], d['loss'], d['kernels']['mrg_seg_reduce_bwd'])
d=json.load(open('gpurun_out/r5y/bench.json')); print('full', d['ms_per_step'], d['loss'], {k: (v['ms_total'], v['achieved']) for k, v in d['kernels'].items() if 'seg_reduce_bwd' in k})
PY
