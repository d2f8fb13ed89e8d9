#!/bin/bash
# round 4, call d: one CU's timeline and the chip-wide phase occupancy of the shipped row GEMM, plain and with the second workgroup of a CU started late
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 120 tools/labbin/x3s_trace_d16 558771 200 200 > $O/timeline.txt 2>&1
timeout -k 5 120 tools/labbin/x3s_trace_d16_stag4 558771 200 200 > $O/timeline_stag4.txt 2>&1
cat $O/timeline.txt $O/timeline_stag4.txt
