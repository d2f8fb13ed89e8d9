#!/bin/bash
# round 4, call y: torch (non-library) kernels left in the headline step, by aten op (tools/torch_ops_profile.py)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4y
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/torch_ops_profile.py > $O/torch_ops.txt 2>&1 || { tail -30 $O/torch_ops.txt; exit 1; }
head -70 $O/torch_ops.txt
