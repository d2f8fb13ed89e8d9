#!/bin/bash
# round 4, call h: GPU tests of the nets + the default bench (caller_reference leg) + CompGCN workload x3 + the full-graph CPU oracle step
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4h
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nets_gpu.py -x -q -m gpu > $O/pytest_nets.txt 2>&1 || { tail -40 $O/pytest_nets.txt; exit 1; }
tail -3 $O/pytest_nets.txt
python bench.py --steps 10 --warmup 3 --no-c5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json
d=json.load(open("gpurun_out/r4h/bench.json"))
print({k:d[k] for k in ("value","ms_per_step","loss")}, d.get("caller_reference"), d["roofline"]["frac"], d["roofline"].get("traffic_source"))
PY
for fn in sub mul ccorr; do python bench.py --workload compgcn_fb15k237 --comp-fn $fn --steps 10 --warmup 3 > $O/bench_compgcn_$fn.json 2> $O/bench_compgcn_$fn.err || { tail -30 $O/bench_compgcn_$fn.err; exit 1; }; done
python - <<'PY'
import json
for fn in ("sub","mul","ccorr"):
    d=json.load(open(f"gpurun_out/r4h/bench_compgcn_{fn}.json"))
    print(fn, {k:d[k] for k in ("value","ms_per_step","loss")}, d.get("ccorr_kernel"), d["roofline"]["kernel"], d["roofline"]["frac"])
PY
python tools/cpu_full_graph.py --out $O/cpu_full_graph.json > $O/cpu_full_graph.txt 2>&1 || { tail -20 $O/cpu_full_graph.txt; exit 1; }
tail -2 $O/cpu_full_graph.txt
