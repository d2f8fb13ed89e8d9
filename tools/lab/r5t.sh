#!/bin/bash
# round 4, call 5t: per-wave phase stamps of the weight gradient's 16-row tile (tools/wgrad_trace_lab.hip)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5t
mkdir -p $O
cd $GRAFT_REPO_ROOT
for shape in "558771 200 200 200" "558771 200 0 200" "558771 100 0 100"; do
  timeout -k 5 120 tools/labbin/wgrad_trace $shape > "$O/trace_$(echo $shape | tr ' ' '_').txt" 2>&1
  cat "$O/trace_$(echo $shape | tr ' ' '_').txt"
done
