#!/bin/bash
# round 4, call 5b: torch (non-library) kernels left in the c5_fixed_cell step
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5b
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/torch_ops_profile.py --workload c5_fixed_cell > $O/torch_ops_c5.txt 2>&1 || { tail -30 $O/torch_ops_c5.txt; exit 1; }
head -45 $O/torch_ops_c5.txt
