#!/bin/bash
# round 4, call 5h: weight gradient with a column block's tiles dealt evenly over its waves: bit identity, timing A/B
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5h
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "weight_gradient or wgrad or linear" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1 || { tail -20 $O/wgrad_probe.txt; exit 1; }
cat $O/wgrad_probe.txt
