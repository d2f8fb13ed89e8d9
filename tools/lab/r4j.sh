#!/bin/bash
# round 4, call j: full-size config tests with the mask replay
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4j
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_configs_gpu.py -x -q -m gpu -s -k "c2 or c3 or c4" > $O/pytest_configs.txt 2>&1 || { tail -60 $O/pytest_configs.txt; exit 1; }
grep -E "mask replay|passed|failed|worst relative" $O/pytest_configs.txt
