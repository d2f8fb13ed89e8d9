#!/bin/bash
# round 4, call u: the same probe at 1 / 2 / 3 segment streams (the 36 ms launch is not a concurrency effect)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4u
mkdir -p $O
cd $GRAFT_REPO_ROOT
for ss in 1 2 3; do
  echo "== MRG_SEGMENT_STREAMS=$ss" >> $O/c5_probe.txt
  MRG_SEGMENT_STREAMS=$ss python tools/c5_probe.py >> $O/c5_probe.txt 2>&1 || { tail -20 $O/c5_probe.txt; exit 1; }
done
cat $O/c5_probe.txt
