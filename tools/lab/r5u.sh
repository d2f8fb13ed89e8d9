#!/bin/bash
# round 4, call 5u: weight gradient with the second wave of each SIMD splitting first and multiplying after: bit identity, timing, stamps, bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5u
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "weight_gradient or wgrad or linear" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1 || { tail -20 $O/wgrad_probe.txt; exit 1; }
cat $O/wgrad_probe.txt
timeout -k 5 120 tools/labbin/wgrad_trace 558771 200 200 200 > $O/trace_dual.txt 2>&1; cat $O/trace_dual.txt
python bench.py --steps 10 --warmup 3 --no-c5 --no-cpu-baseline --no-caller-leg --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r5u/bench.json')); print(d['ms_per_step'], d['loss'], d['kernel_families']['weight_gradient [wgrad_x3v_k]'])"
