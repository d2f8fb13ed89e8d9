#!/bin/bash
# round 4, call 5c: fixed-genotype cell with reader aliases (Fan), fused BN + ReLU after the concat Linear, the scorer's output gradient
# transposed with its activation derivative in one pass: parity suites, C5 / C1 bench lines, torch-op profile of the C5 step
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5c
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_nets_gpu.py tests/test_configs_gpu.py tests/test_ops_gpu.py tests/test_dataprep_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python bench.py --workload c5_fixed_cell > $O/bench_c5.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 1; }
python bench.py --workload fb15k237_fixed_d64 > $O/bench_c1.json 2> $O/bench_c1.err || { tail -20 $O/bench_c1.err; exit 1; }
python - <<'PY'
import json
for f in ("bench_c5.json", "bench_c1.json"):
    d = json.load(open("gpurun_out/r5c/" + f))
    print(f, d["ms_per_step"], d["value"], d["loss"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
python tools/torch_ops_profile.py --workload c5_fixed_cell > $O/torch_ops_c5.txt 2>&1 || { tail -30 $O/torch_ops_c5.txt; exit 1; }
head -14 $O/torch_ops_c5.txt
