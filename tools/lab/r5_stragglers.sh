#!/bin/bash
# Round 5 lab: the HBM stragglers A/B (gate kernels two rows per trip, aggregator backward in destination order) at the C5 and FB shapes.
mkdir -p gpurun_out/r5
for shape in c5 fb; do
  for cfg in "MRG_GATE_RPT=1 MRG_SEG_BWD_ORDERED=0" "MRG_GATE_RPT=2 MRG_SEG_BWD_ORDERED=1"; do
    echo "== $shape $cfg"
    env $cfg timeout -k 10 300 python tools/kbench.py --shape $shape --only gate,seg --reps 7 2>&1 | grep "^gate\|^seg" | cut -c1-160
  done
done
