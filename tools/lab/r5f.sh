#!/bin/bash
# round 4, call 5f: the Python lines that issue the torch (aten) kernels of the headline step
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5f
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/aten_sites.py > $O/aten_sites.txt 2>&1 || { tail -30 $O/aten_sites.txt; exit 1; }
grep -v "amdgpu.ids" $O/aten_sites.txt | head -80
