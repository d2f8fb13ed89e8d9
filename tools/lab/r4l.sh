#!/bin/bash
# round 4, call l: column blocks of four tiles at three waves per SIMD (A read twice) against the shipped seven-tile kernel: timing only
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4l
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in x3s_dbg_0 x3s_nt4_w2 x3s_nt4_w3; do echo -n "$v: " >> $O/nt4.txt; timeout -k 5 60 tools/labbin/$v 558771 200 200 0 >> $O/nt4.txt 2>&1; done; done
for v in x3s_dbg_0 x3s_nt4_w2 x3s_nt4_w3; do echo -n "$v: " >> $O/nt4.txt; timeout -k 5 60 tools/labbin/$v 558771 400 200 0 >> $O/nt4.txt 2>&1; done
cat $O/nt4.txt
