#!/bin/bash
# Same-box A/B of environment switches on ONE command: each configuration is run ROUNDS times, interleaved (A B B A ...), so
# the box's clocks and neighbours hit both sides (cdna_hip_programming.md rule 24).
#
#   tools/lab/ab.sh TAG [--rounds N] [--grep REGEX] 'ENV=.. ENV=..' 'ENV=.. ENV=..' [more configs] -- COMMAND...
#
# With a COMMAND that is a list of bench.py flags (first word starts with "--") every run's ms_per_step is collected and the
# median / min per configuration printed; otherwise the lines matching --grep (default: all) are printed under each run.
#
# Examples (rounds 4 / 5):
#   gate kernels one / two rows per trip, aggregator backward in edge / destination order, at the C5 and FB shapes:
#     tools/lab/ab.sh stragglers --rounds 1 --grep '^gate|^seg' 'MRG_GATE_RPT=1 MRG_SEG_BWD_ORDERED=0' 'MRG_GATE_RPT=2 MRG_SEG_BWD_ORDERED=1' \
#         -- python tools/kbench.py --shape c5 --only gate,seg --reps 7
#   three-waves-per-SIMD row GEMM against the two-wave kernel in the headline step:
#     tools/lab/ab.sh q 'MRG_GEMM_Q=0' 'MRG_GEMM_Q=1' -- --steps 10 --warmup 3 --no-cpu-baseline
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
ROUNDS=3; GREP="."
CFG=()
while [ $# -gt 0 ]; do
  case "$1" in
    --rounds) ROUNDS=$2; shift 2;;
    --grep) GREP=$2; shift 2;;
    --) shift; break;;
    *) CFG+=("$1"); shift;;
  esac
done
first=$1
n=${#CFG[@]}
for ((r = 0; r < ROUNDS; r++)); do
  for ((j = 0; j < n; j++)); do
    i=$(( r % 2 == 0 ? j : n - 1 - j ))
    out="$O/cfg${i}_round${r}.txt"
    echo "== round $r: ${CFG[$i]}"
    if [ "${first#--}" != "$first" ]; then
      env ${CFG[$i]} timeout -k 10 ${LAB_TIMEOUT:-600} python bench.py "$@" > "$out" 2> "$out.err" || { tail -25 "$out.err"; exit 1; }
    else
      env ${CFG[$i]} timeout -k 10 ${LAB_TIMEOUT:-600} "$@" > "$out" 2>&1 || { tail -25 "$out"; exit 1; }
      grep -E "$GREP" "$out" | cut -c1-200
    fi
  done
done
if [ "${first#--}" != "$first" ]; then
  python - "$O" "$n" "$ROUNDS" "${CFG[@]}" <<'PY'
import json, statistics, sys
o, n, rounds, cfg = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:]
for i in range(n):
    ms = [json.loads(open(f"{o}/cfg{i}_round{r}.txt").read().strip().splitlines()[-1])["ms_per_step"] for r in range(rounds)]
    print(f"{cfg[i]:60s} median {statistics.median(ms):9.3f}  min {min(ms):9.3f} ms/step  {ms}")
PY
fi
