#!/bin/bash
# round 4, call 5w: the rebuilt library (lab switches compiled out) -- operator tests and a short bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5w
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -2 $O/pytest.txt
python bench.py --no-cpu-baseline --no-c5 --no-caller-leg --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r5w/bench.json')); print(d['ms_per_step'], d['loss'], d['roofline']['frac'])"
