#!/bin/bash
# round 4, call 5z: rocprofv3 --kernel-trace --stats of the c5_fixed_cell bench line at HEAD
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5z
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc5 -- python3 $R/bench.py --workload c5_fixed_cell --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_rocprof.json 2> $O/bench_c5_rocprof.err || { tail -20 $O/bench_c5_rocprof.err; exit 1; }
find /tmp/pc5 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_c5.csv \;
head -12 $O/kernel_stats_c5.csv | cut -c1-150
python3 -c "
import json; d=json.load(open('$O/bench_c5_rocprof.json')); print(d['ms_per_step'], d['roofline']['kernel'], d['roofline']['us_per_launch'], d['roofline']['launches'])"
