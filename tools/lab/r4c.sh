#!/bin/bash
# round 4, call c: cumulative component removal on the shipped row GEMM with phase stamps, two workgroups per CU and one (extra LDS)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4c
mkdir -p $O
cd $GRAFT_REPO_ROOT
for x in 0 40000; do for d in 16 20 22 30 94 222; do
  echo "=== dbg=$d lds_extra=$x" >> $O/ablate.txt
  MRG_X3S_LDS_EXTRA=$x timeout -k 5 120 tools/labbin/x3s_trace_d$d 558771 200 200 2>&1 | grep -E "x3s trace|clock|prologue|k-loop|epilogue|slab  [4-9]|CUs seen" >> $O/ablate.txt
done; done
cat $O/ablate.txt
