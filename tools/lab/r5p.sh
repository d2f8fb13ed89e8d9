#!/bin/bash
# round 4, call 5p: tests of module_linear and the deferred BatchNorm counters
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5p
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "module_linear or deferred or act_grad or wide_short" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
