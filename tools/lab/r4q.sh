#!/bin/bash
# round 4, call q: host profile of the launch-bound 300-edge search step (cProfile), HEAD
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/host_profile.py > $O/host_profile_300.txt 2>&1
head -70 $O/host_profile_300.txt
