#!/bin/bash
# round 4, call g: branch-free steady-state slabs (MRG_X3S_STEADY=1) vs round 3's run-time form: timing A/B, stamps, bit identity
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4g
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for v in st0 st1; do for m in 0; do timeout -k 5 60 tools/labbin/x3s_$v 558771 200 200 $m >> $O/ab.txt 2>&1; done; done; done
for v in st0 st1 st0 st1; do timeout -k 5 60 tools/labbin/x3s_$v 558771 400 200 0 >> $O/ab.txt 2>&1; done
for v in st0 st1 st0 st1; do timeout -k 5 60 tools/labbin/x3s_$v 272115 200 200 0 >> $O/ab.txt 2>&1; done
timeout -k 5 120 tools/labbin/x3s_trace_st1 558771 200 200 2>&1 | grep -v "^   start" > $O/trace_st1.txt
MRG_X3S_LDS_EXTRA=40000 timeout -k 5 120 tools/labbin/x3s_trace_st1 558771 200 200 2>&1 | grep -v "^   start" > $O/trace_st1_1wg.txt
MRG_X3S_LDS_EXTRA=40000 timeout -k 5 120 tools/labbin/x3s_trace_st1_d222 558771 200 200 2>&1 | grep -v "^   start" > $O/trace_st1_d222_1wg.txt
timeout -k 5 200 tools/labbin/gemm_x3_lab_st1 558771 200 0 200 > $O/lab_558k.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_st1 272115 200 200 200 > $O/lab_272k_dual.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_st1 70001 128 0 100 > $O/lab_70k.txt 2>&1
timeout -k 5 200 tools/labbin/gemm_x3_lab_st1 3000 96 0 96 > $O/lab_3k.txt 2>&1
cat $O/ab.txt; head -22 $O/trace_st1.txt; grep -E "k-loop|slab  [4-7]|clock" $O/trace_st1_1wg.txt $O/trace_st1_d222_1wg.txt; grep -E "x3s vs x3|x3s LDS|differ" $O/lab_*.txt
