#!/bin/bash
# Round 5 lab: the gate kernels with tied operands (what the C5 fixed cell launches) at 1 / 2 rows per trip, next to the streaming calibration kernel
for cfg in "MRG_GATE_RPT=1" "MRG_GATE_RPT=2"; do
  echo "== c5 $cfg"
  env $cfg timeout -k 10 300 python tools/kbench.py --shape c5 --only gate,compose --reps 7 2>&1 | grep "^gate\|^compose" | cut -c1-160
done
