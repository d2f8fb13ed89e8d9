#!/bin/bash
# round 4, call s: the C5-shaped input-gradient product with the eight-tile block on and off
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4s
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/wide8_probe.py > $O/wide8_probe.txt 2>&1 || { tail -20 $O/wide8_probe.txt; exit 1; }
cat $O/wide8_probe.txt
