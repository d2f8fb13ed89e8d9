#!/bin/bash
# round 4, call 5m: what the driver runs at round end -- the full GPU suite, smoke(), the default bench line
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5m
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { tail -20 $O/smoke.txt; exit 1; }
tail -2 $O/smoke.txt
T0=$(date +%s)
python bench.py > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
echo "default bench.py wall clock: $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r5m/bench.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["loss"])
PY
