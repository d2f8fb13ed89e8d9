#!/bin/bash
# round 4, call 5l: BatchNorm step counters bumped by one launch per forward: net / config / dist tests, sampled and headline steps
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5l
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nets_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for w in fb15k237_supernet_300 fb15k237_supernet_30k fb15k237_supernet_full; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-c5 --no-cpu-baseline --no-caller-leg --no-exact-f32-leg > $O/bench_$w.json 2> $O/bench_$w.err || { tail -30 $O/bench_$w.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5l/bench*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["loss"])
PY
