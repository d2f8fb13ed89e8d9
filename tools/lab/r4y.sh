#!/bin/bash
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4y
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/torch_ops_profile.py > $O/torch_ops.txt 2>&1 || { tail -30 $O/torch_ops.txt; exit 1; }
head -70 $O/torch_ops.txt
