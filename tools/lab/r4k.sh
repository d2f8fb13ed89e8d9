#!/bin/bash
# round 4, call k: after the Candidate / Link refactor -- operator and net tests, the MixedOp fuzzer, a short bench
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4k
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_dist_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -60 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/fuzz_mixed.py > $O/fuzz_mixed.txt 2>&1 || { tail -30 $O/fuzz_mixed.txt; exit 1; }
tail -3 $O/fuzz_mixed.txt
python bench.py --steps 10 --warmup 3 --no-c5 --no-cpu-baseline --no-caller-leg --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4k/bench.json')); print(d['value'], d['ms_per_step'], d['loss'])"
python -m pytest tests/test_compgcn_gpu.py tests/test_plans_gpu.py tests/test_dataprep_gpu.py -x -q -m gpu > $O/pytest2.txt 2>&1 || { tail -60 $O/pytest2.txt; exit 1; }
tail -3 $O/pytest2.txt
python bench.py --workload compgcn_fb15k237 --comp-fn ccorr --steps 10 --warmup 3 > $O/bench_ccorr.json 2> $O/bench_ccorr.err || { tail -30 $O/bench_ccorr.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4k/bench_ccorr.json')); print(d['value'], d['ms_per_step'], d['loss'], d.get('ccorr_kernel'))"
