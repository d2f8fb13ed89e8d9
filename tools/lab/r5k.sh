#!/bin/bash
# round 4, call 5k: the Python lines that issue the torch (aten) kernels of one rank's step of the 8-way sharded supernet
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5k
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/aten_sites.py --rehearse-shard 3/8 > $O/aten_sites_shard.txt 2>&1 || { tail -30 $O/aten_sites_shard.txt; exit 1; }
grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/aten_sites_shard.txt | head -75
