#!/bin/bash
# The profiling passes of one bench.py configuration, each in its own rocprofv3 run (counters never share a run with a
# sys / runtime trace; the program itself follows "--"):
#
#   tools/lab/profile.sh TAG PASSES [bench.py flags...]        PASSES = comma list of  stats,stats1,traffic,sq,census
#
#   stats    rocprofv3 --kernel-trace --stats of the command as benched (multi-stream)   -> gpurun_out/TAG/kernel_stats.csv, bench.json
#   stats1   the same with MRG_MIXED_STREAMS=1 MRG_SEGMENT_STREAMS=1 (durations without CU sharing) -> kernel_stats_single_stream.csv
#   traffic  --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each (MI355X_MICROARCH.md's rocprofv3 section) -> pmc_fetch/, pmc_write/
#            (tools/traffic_from_pmc.py turns them into bytes per launch with the guide's unit and gfx950 corrections)
#   sq       the SQ counters (MFMA busy, wave cycles, waits, LDS conflicts): tools/pmc_sq.sh      -> sq_summary.txt
#   census   kernels per step by name (calls / step, average, ms / step) from the stats CSV (needs the stats pass in the same call)
#
# The stats passes run the flags as given (default: bench.py's own defaults, CPU baseline included, extra legs off); the counter
# passes add --steps 1 --warmup 1 --no-cpu-baseline.
# Examples: the round's profile set     tools/lab/profile.sh r5prof stats,stats1,traffic,sq
#           C5                          tools/lab/profile.sh r5prof_c5 stats,sq --workload c5_fixed_cell
#           the replayed sampled step   tools/lab/profile.sh r5static stats,census --workload fb15k237_supernet_300 --resample --static-step --hip-graph --steps 50 --warmup 5 --no-cpu-baseline
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
TAG=$1; PASSES=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
LEGS="--no-exact-f32-leg --no-caller-leg"
has() { case ",$PASSES," in *",$1,"*) return 0;; esac; return 1; }
if has stats; then
  rm -rf /tmp/prof_a
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_a -- python3 $R/bench.py $LEGS "$@" > $O/bench.json 2> $O/bench.err || { tail -8 $O/bench.err; exit 1; }
  find /tmp/prof_a -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
  echo "stats done: $(grep timed $O/bench.err)"
fi
if has stats1; then
  rm -rf /tmp/prof_b
  MRG_MIXED_STREAMS=1 MRG_SEGMENT_STREAMS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python3 $R/bench.py $LEGS --no-cpu-baseline "$@" > $O/bench_single_stream.json 2> $O/bench_single_stream.err || { tail -8 $O/bench_single_stream.err; exit 1; }
  find /tmp/prof_b -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats_single_stream.csv \;
  echo "stats1 done: $(grep timed $O/bench_single_stream.err)"
fi
if has traffic; then
  for c in FETCH_SIZE:pmc_fetch WRITE_SIZE:pmc_write; do
    ctr=${c%%:*}; d=${c#*:}
    rm -rf $O/$d; mkdir -p $O/$d
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/$d -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline $LEGS "$@" > $O/$d.json 2> $O/$d.err || { tail -8 $O/$d.err; exit 1; }
    find $O/$d -name "*kernel_trace.csv" -delete; find $O/$d -name "*agent_info.csv" -delete
    echo "$ctr pass done"
  done
fi
if has sq; then bash $R/tools/pmc_sq.sh "$TAG" "$@" || exit 1; fi
if has census; then
  python3 - "$O/kernel_stats.csv" "$O/bench.json" <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
replay = str(d.get("config", {}).get("launch", "")).startswith("hip graph")
steps = d["steps"] + (0 if replay else d["warmup"])      # a captured step's warm-ups run eagerly and are in the trace too: an upper bound per step
calls = sum(int(r["Calls"]) for r in rows); tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{calls / steps:.0f} kernels/step over {steps} steps, {len(rows)} distinct, {tot / 1e6 / steps:.2f} ms/step of kernel time")
for r in rows[:40]:
    print(f"  {r['Name'][:80]:80s} {int(r['Calls']) / steps:8.1f} {float(r['AverageNs']) / 1e3:8.1f} us {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step")
PY
fi
du -sh $O; ls $O
