#!/bin/bash
# Kernel-trace profile of one bench.py run on the GPU box; keeps only the per-kernel statistics
# (the full trace is far beyond what gpurun copies back).  usage: tools/profile_bench.sh TAG [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py "$@" > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}.err
rc=$?
mkdir -p $R/gpurun_out/prof_$TAG
find /tmp/prof_$TAG -name "*stats*.csv" -exec cp {} $R/gpurun_out/prof_$TAG/ \;
grep timed $R/gpurun_out/${TAG}.err
exit $rc
