#!/bin/bash
# round 4, call 5v: what would HALF the matrix instructions buy the row GEMM?  (lab switch 1024: the three 2^-16 terms left out, wrong
# results, timing only; 1088 = that without the A split's arithmetic as well)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5v
mkdir -p $O
cd $GRAFT_REPO_ROOT
for shape in "272115 200 200" "558771 200 200" "558771 400 200" "2000000 256 256"; do
  for d in 0 1024 1088 0 1024; do timeout -k 5 60 tools/labbin/x3s_dbg_$d $shape; done
done > $O/three_terms.txt 2>&1
cat $O/three_terms.txt
