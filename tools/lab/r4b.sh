#!/bin/bash
# round 4, call b: pipelined slab schedule (MRG_X3S_PIPE=1) vs round 3's: stamps, A/B timing, bit identity with the one-wave kernel
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 120 tools/labbin/x3s_trace_p1 558771 200 200 >> $O/trace.txt 2>&1
timeout -k 5 120 tools/labbin/x3s_trace_p1 558771 400 200 >> $O/trace.txt 2>&1
for v in p0 p1 p0 p1; do echo "== $v 558k" >> $O/lab.txt; timeout -k 5 200 tools/labbin/gemm_x3_lab_$v 558771 200 0 200 >> $O/lab.txt 2>&1; done
for v in p0 p1; do echo "== $v 272k dual" >> $O/lab.txt; timeout -k 5 200 tools/labbin/gemm_x3_lab_$v 272115 200 200 200 >> $O/lab.txt 2>&1; done
for v in p1; do echo "== $v 70k N=128" >> $O/lab.txt; timeout -k 5 200 tools/labbin/gemm_x3_lab_$v 70000 128 0 128 >> $O/lab.txt 2>&1; done
cat $O/trace.txt; grep -E "==|x3s|x3 \(gemm only, acc|float64" $O/lab.txt
