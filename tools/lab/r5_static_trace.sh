#!/bin/bash
# Round 5 lab: kernel census of the replayed static search step (graph_batch_size 300): which kernels, how many, how long
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
O=$R/gpurun_out/r5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ps
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps -- python3 $R/bench.py --workload ${1:-fb15k237_supernet_300} --resample --static-step --hip-graph --steps 50 --warmup 3 --no-cpu-baseline --no-c5 --no-exact-f32-leg --no-caller-leg > $O/static_trace_bench.json 2> $O/static_trace_bench.err || exit 1
find /tmp/ps -name "*kernel_stats.csv" -exec cp {} $O/static_kernel_stats.csv \;
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/static_kernel_stats.csv")))
d = json.load(open("$O/static_trace_bench.json"))
steps = 50 + 3 + 3          # timed + warmup + capture warm-ups / instrumented (approximate: per-step figures below use the replay count)
tot = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
print("ms_per_step", d["ms_per_step"], "| all kernels:", calls, "calls", round(tot / 1e6, 2), "ms in the whole run")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:40]:
    print("%-64s calls %6s  total %8.2f ms  avg %7.2f us" % (r["Name"][:64], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
