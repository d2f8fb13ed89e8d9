#!/bin/bash
# round 4, call 5e: torch (non-library) kernels in one rank's step of the 8-way sharded supernet (timing-only rehearsal)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5e
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/torch_ops_profile.py --rehearse-shard 3/8 > $O/torch_ops_shard.txt 2>&1 || { tail -30 $O/torch_ops_shard.txt; exit 1; }
grep -v "Warning\|_warn_once\|amdgpu.ids" $O/torch_ops_shard.txt | head -50
