#!/bin/bash
# round 4, call 5a: the full-graph supernet step eager vs replayed from a HIP graph, same box
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5a
mkdir -p $O
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-c5 --no-exact-f32-leg --no-caller-leg --steps 10 --warmup 3"
python bench.py $F > $O/eager.json 2> $O/eager.err || { tail -20 $O/eager.err; exit 1; }
python bench.py $F --hip-graph > $O/graph.json 2> $O/graph.err || { tail -20 $O/graph.err; exit 1; }
python - <<'PY'
import json
for f in ("eager.json", "graph.json"):
    d = json.load(open("gpurun_out/r5a/" + f))
    print(f, d["ms_per_step"], d["value"], d["loss"], d["config"]["launch"])
PY
