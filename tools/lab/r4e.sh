#!/bin/bash
# round 4, call e: transposed accumulators (16-byte epilogue accesses) in the LDS-weight row GEMM: bit identity, timing, stamps
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4e
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 120 tools/labbin/x3s_trace_tr 558771 200 200 2>&1 | grep -v "^   start" > $O/trace_tr.txt
timeout -k 5 200 tools/labbin/gemm_x3_lab_tr 558771 200 0 200 > $O/lab_558k.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_tr 272115 200 200 200 > $O/lab_272k_dual.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_tr 70001 128 0 100 > $O/lab_70k.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_tr 14541 200 0 200 > $O/lab_14k.txt 2>&1
cat $O/trace_tr.txt; grep -E "x3s|differ|float64|x3 \(gemm only, acc" $O/lab_*.txt
