#!/bin/bash
# round 4, call p: the strip's stores PACED through the k-loop (emulation: same addresses and count, garbage values) against the burst epilogue
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4p
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do for v in x3s_dbg_0 x3s_dbg_1 x3s_dbg_513; do echo -n "$v: " >> $O/paced.txt; timeout -k 5 60 tools/labbin/$v 558771 200 200 0 >> $O/paced.txt 2>&1; done; done
for v in x3s_dbg_0 x3s_dbg_1 x3s_dbg_513; do echo -n "$v: " >> $O/paced.txt; timeout -k 5 60 tools/labbin/$v 558771 400 200 0 >> $O/paced.txt 2>&1; done
timeout -k 5 120 tools/labbin/x3s_trace_529 558771 200 200 2>&1 | grep -v "^   start" > $O/trace_paced.txt
cat $O/paced.txt; head -24 $O/trace_paced.txt
