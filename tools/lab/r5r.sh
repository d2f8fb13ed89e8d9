#!/bin/bash
# round 4, call 5r: long fuzz runs at HEAD with new seeds (300 MixedOp cases, 300 operator-path cases)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python tools/fuzz_mixed.py 300 4 > $O/fuzz_mixed_300.txt 2>&1 || { tail -20 $O/fuzz_mixed_300.txt; exit 1; }
tail -2 $O/fuzz_mixed_300.txt
timeout -k 10 500 python tools/fuzz_paths.py 300 4 > $O/fuzz_paths_300.txt 2>&1 || { tail -20 $O/fuzz_paths_300.txt; exit 1; }
tail -2 $O/fuzz_paths_300.txt
