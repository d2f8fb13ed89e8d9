#!/bin/bash
# round 4, call z: node-level nn.Linear modules on the library's row GEMM: parity suites + the headline bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4z
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nets_gpu.py tests/test_configs_gpu.py tests/test_dist_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python bench.py --no-cpu-baseline --no-c5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
MRG_NODE_LINEAR=0 python bench.py --no-cpu-baseline --no-c5 --no-exact-f32-leg --no-caller-leg > $O/bench_torch_linear.json 2> $O/bench2.err || { tail -20 $O/bench2.err; exit 1; }
python - <<'PY'
import json
for f in ("bench.json", "bench_torch_linear.json"):
    d = json.load(open("gpurun_out/r4z/" + f))
    print(f, d["ms_per_step"], d["value"], d["loss"], d["roofline"]["frac"])
PY
