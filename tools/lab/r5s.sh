#!/bin/bash
# round 4, call 5s: which tensor carries the 1e-4-class differences of the three fuzz cases of seed 4 that exceed 5e-5
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5s
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/fuzz_mixed.py 120 4 > $O/fuzz_mixed_120.txt 2>&1 || true
grep -v " ok " $O/fuzz_mixed_120.txt | grep -v amdgpu
timeout -k 10 500 python tools/fuzz_paths.py 300 4 > $O/fuzz_paths_300.txt 2>&1 || { tail -20 $O/fuzz_paths_300.txt; exit 1; }
tail -2 $O/fuzz_paths_300.txt
