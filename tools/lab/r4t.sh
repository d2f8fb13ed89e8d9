#!/bin/bash
# round 4, call t: per-launch times of the row-GEMM entry points inside one c5_fixed_cell step, eight-tile block off / on (tools/c5_probe.py)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4t
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/c5_probe.py > $O/c5_probe.txt 2>&1 || { tail -20 $O/c5_probe.txt; exit 1; }
cat $O/c5_probe.txt
