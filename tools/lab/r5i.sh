#!/bin/bash
# round 4, call 5i: is the weight gradient bound by the latency of its one-tile-deep prefetch?  every prefetch re-reads tile 0 (MRG_WGRAD_LAB=1)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5i
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/wgrad_probe.py > $O/wgrad_probe.txt 2>&1 || { tail -20 $O/wgrad_probe.txt; exit 1; }
MRG_WGRAD_LAB=1 python tools/wgrad_probe.py > $O/wgrad_probe_lab1.txt 2>&1 || { tail -20 $O/wgrad_probe_lab1.txt; exit 1; }
echo "== default"; cat $O/wgrad_probe.txt; echo "== every prefetch reads tile 0"; cat $O/wgrad_probe_lab1.txt
