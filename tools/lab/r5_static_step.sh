#!/bin/bash
# Round 5 lab: the reference's sampled search step (graph_batch_size 300 / 30 000) with a new draw every step:
#   eager with host reads (round 4's --resample), eager static (no host read), static + captured whole-step HIP graph
mkdir -p gpurun_out/r5
for wl in fb15k237_supernet_300 fb15k237_supernet_30k; do
  for mode in "--resample" "--resample --static-step" "--resample --static-step --hip-graph"; do
    echo "== $wl $mode"
    timeout -k 10 240 python bench.py --workload $wl $mode --steps 30 --warmup 5 --no-cpu-baseline --no-c5 --no-exact-f32-leg --no-caller-leg 2> gpurun_out/r5/static_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], 'ms/step', d['value'], 'M edges/s', d['config']['launch'], 'loss', d['loss'])" || tail -5 gpurun_out/r5/static_err.txt
  done
done
