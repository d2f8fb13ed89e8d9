#!/bin/bash
# Lab: per-dispatch durations of selected kernels in one bench run (single-stream so durations are not shared).
# usage: [STREAMS=multi] tools/trace_kernels.sh TAG 'regex' [bench args...]   (columns: index, start ns, duration, grid, kernel, previous kernel)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PAT=$2; shift 2
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tk
if [ "${STREAMS:-single}" = single ]; then export MRG_MIXED_STREAMS=1 MRG_SEGMENT_STREAMS=1; fi
rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exact-f32-leg --no-c5 "$@" > $O/bench.json 2> $O/bench.err || exit 1
f=$(find /tmp/tk -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$PAT" > $O/trace.txt <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
pat = re.compile(sys.argv[2])
for i, r in enumerate(rows):
    if pat.search(r["Kernel_Name"]):
        prev = rows[i - 1]["Kernel_Name"][:60] if i else ""
        print(f'{i:6d} {int(r["Start_Timestamp"])} {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us grid {r.get("Grid_Size", r.get("Grid_Size_X", "?")):>9s} {r["Kernel_Name"][:70]:70s} after {prev}')
PY
wc -l $O/trace.txt
