#!/bin/bash
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4t
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/c5_probe.py > $O/c5_probe.txt 2>&1 || { tail -20 $O/c5_probe.txt; exit 1; }
cat $O/c5_probe.txt
