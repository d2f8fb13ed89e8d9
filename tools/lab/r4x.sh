#!/bin/bash
# round 4, call x: the ring-of-two row GEMM at SEVEN tiles (N = 200) against the ring-of-three default, lab timing + bit identity
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4x
mkdir -p $O
cd $GRAFT_REPO_ROOT
for rows in 272115 558771 2000000; do
  MRG_LAB_RING2=1 timeout -k 5 300 tools/labbin/gemm_x3_lab $rows 200 0 200 5 > $O/lab_${rows}_200.txt 2>&1
  echo "== rows $rows"; grep -E "x3s8|x3s \(N|x3s accumulate|x3s LDS-B" $O/lab_${rows}_200.txt
done
