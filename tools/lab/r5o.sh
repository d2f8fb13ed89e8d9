#!/bin/bash
# round 4, call 5o: Infinity Cache probe -- streaming rate of a three-buffer kernel against the buffer size, and of a producer -> consumer pair
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5o
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/mall_probe.py > $O/mall_probe.txt 2>&1 || { tail -20 $O/mall_probe.txt; exit 1; }
grep -v amdgpu.ids $O/mall_probe.txt
