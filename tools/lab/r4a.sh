#!/bin/bash
# round 4, call a: phase stamps of the shipped row GEMM kernel
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4a
mkdir -p $O
cd $GRAFT_REPO_ROOT
for r in 272115 558771; do timeout -k 5 120 tools/labbin/x3s_trace $r 200 200 >> $O/trace.txt 2>&1; done
timeout -k 5 120 tools/labbin/x3s_trace 558771 400 200 >> $O/trace.txt 2>&1
cat $O/trace.txt
