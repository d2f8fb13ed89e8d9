#!/bin/bash
# round 4, call w: full GPU suite and the C5 / headline bench lines after the reroute
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4w
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python bench.py --workload c5_fixed_cell > $O/bench_c5.json 2> $O/bench_c5.err || { tail -20 $O/bench_c5.err; exit 1; }
cat $O/bench_c5.json
python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
python bench.py --workload fixed_d64 > $O/bench_fixed.json 2> $O/bench_fixed.err || { tail -20 $O/bench_fixed.err; exit 1; }
cat $O/bench_fixed.json
