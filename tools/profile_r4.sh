#!/bin/bash
# All profiling passes of one bench.py configuration on the GPU box (run through gpurun):
#   1. rocprofv3 --kernel-trace --stats of the default (multi-stream) command       -> TAG/kernel_stats.csv, TAG/bench.json
#   2. the same with MRG_MIXED_STREAMS=1 MRG_SEGMENT_STREAMS=1 (kernel durations without CU sharing) -> TAG/kernel_stats_single_stream.csv
#   3. --pmc FETCH_SIZE   4. --pmc WRITE_SIZE  (separate passes, kernel-trace only)  -> TAG/pmc_fetch, TAG/pmc_write
# usage: tools/profile_r4.sh TAG [passes: default 1234]      (round 4: the extra legs of the default bench are switched off in the counter passes)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PASSES=${2:-1234}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
if [[ $PASSES == *1* ]]; then
  rm -rf /tmp/p1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-exact-f32-leg --no-caller-leg > $O/bench.json 2> $O/bench.err || exit 1
  find /tmp/p1 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  echo "pass 1 done"; grep timed $O/bench.err
fi
if [[ $PASSES == *2* ]]; then
  rm -rf /tmp/p2
  MRG_MIXED_STREAMS=1 MRG_SEGMENT_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exact-f32-leg --no-caller-leg > $O/bench_single_stream.json 2> $O/bench_single_stream.err || exit 1
  find /tmp/p2 -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_single_stream.csv \;
  echo "pass 2 done"; grep timed $O/bench_single_stream.err
fi
if [[ $PASSES == *3* ]]; then
  rm -rf $O/pmc_fetch; mkdir -p $O/pmc_fetch
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32-leg --no-caller-leg > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 1
  find $O/pmc_fetch -name "*kernel_trace.csv" -delete; find $O/pmc_fetch -name "*agent_info.csv" -delete
  echo "pass 3 done"
fi
if [[ $PASSES == *4* ]]; then
  rm -rf $O/pmc_write; mkdir -p $O/pmc_write
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32-leg --no-caller-leg > $O/pmc_write.json 2> $O/pmc_write.err || exit 1
  find $O/pmc_write -name "*kernel_trace.csv" -delete; find $O/pmc_write -name "*agent_info.csv" -delete
  echo "pass 4 done"
fi
du -sh $O
