#!/bin/bash
# round 4, call 5x: compose backward with two vectors per trip and all loads first: operator tests, probe, C5 bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5x
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python tools/compose_probe.py > $O/compose_probe.txt 2>&1 || { tail -20 $O/compose_probe.txt; exit 1; }
grep -v amdgpu $O/compose_probe.txt
python bench.py --workload c5_fixed_cell --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r5x/bench_c5.json')); print(d['ms_per_step'], d['loss'], d['kernels']['mrg_compose_bwd'])"
