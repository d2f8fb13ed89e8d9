#!/bin/bash
# round 4, call o: same-box A/B of the launch-bound sampled steps, round-3 tree (gpurun_in/r3tree) against HEAD
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4o
mkdir -p $O
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for w in fb15k237_supernet_30k fb15k237_supernet_300; do
  (cd gpurun_in/r3tree && python bench.py --workload $w --steps 30 --warmup 5 --no-c5 --no-cpu-baseline) > $O/r3_${w}_$i.json 2> $O/r3_${w}_$i.err
  python bench.py --workload $w --steps 30 --warmup 5 --no-c5 --no-cpu-baseline --no-caller-leg > $O/head_${w}_$i.json 2> $O/head_${w}_$i.err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4o/*.json")):
    d=json.load(open(f)); print(f.split("/")[-1], d["ms_per_step"])
PY
