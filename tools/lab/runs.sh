#!/bin/bash
# One gpurun call = a list of labelled commands run one after the other on the same box, each under its own timeout, each with
# its stdout / stderr kept under gpurun_out/TAG/ and the line(s) worth reading printed at the end.
#
#   tools/lab/runs.sh TAG [--tests 'PYTEST ARGS'] [--grep REGEX] LABEL::'COMMAND' [LABEL::'COMMAND' ...]
#
# COMMAND is a shell word list run from the repo root; a bare list of bench.py flags (first word starts with "--" or is empty)
# is run as `python bench.py <flags>` and its JSON line is summarised (ms/step, value, loss, launch).  --tests runs
# `python -m pytest <args>` first and stops the call if it fails (one process, as the GPU box requires).  The call stops at the
# first command that fails or times out (no further GPU step after a kill).
#
# Examples (each was one lab call of rounds 4 / 5):
#   the static sampled step, three modes x two sizes:
#     tools/lab/runs.sh static 300::'--workload fb15k237_supernet_300 --resample --no-cpu-baseline --steps 30 --warmup 5' \
#         300s::'--workload fb15k237_supernet_300 --resample --static-step --no-cpu-baseline --steps 30 --warmup 5' ...
#   the round-end set:   tools/lab/runs.sh end --tests 'tests -m gpu -q -x' smoke::'python -c "import __graft_entry__ as g; g.smoke()"' bench::''
#   soak:                tools/lab/runs.sh soak full::'python tools/soak.py' 30k::'python tools/soak.py --workload fb15k237_supernet_30k --resample'
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p "$O"
GREP=""
while [ $# -gt 0 ]; do
  case "$1" in
    --tests) shift; timeout -k 10 ${LAB_TEST_TIMEOUT:-900} bash -c "python -m pytest $1 -p no:cacheprovider" > "$O/tests.log" 2>&1; rc=$?; tail -4 "$O/tests.log"; [ $rc -eq 0 ] || exit $rc; shift;;
    --grep) shift; GREP=$1; shift;;
    *) break;;
  esac
done
for spec in "$@"; do
  label=${spec%%::*}; cmd=${spec#*::}
  first=${cmd%% *}
  echo "== $label: $cmd"
  if [ -z "$cmd" ] || [ "${first#--}" != "$first" ]; then
    timeout -k 10 ${LAB_TIMEOUT:-600} python bench.py $cmd > "$O/$label.json" 2> "$O/$label.err"; rc=$?
    if [ $rc -ne 0 ]; then tail -25 "$O/$label.err"; exit $rc; fi
    python - "$O/$label.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
c = d.get("config", {})
val = "value=null (timing only)" if d["value"] is None else f"{d['value']:.4f} {d['unit']}"
print(f"   {d['ms_per_step']:.3f} ms/step  {val}  launch={c.get('launch')}  loss={d.get('loss')}"
      + (f"  roofline.frac={d['roofline']['frac']:.3f}" if d.get("roofline") else "")
      + "".join(f"  {k}={v.get('ms_per_step')}" for k, v in d.items() if k.startswith("caller_") and isinstance(v, dict)))
PY
  else
    timeout -k 10 ${LAB_TIMEOUT:-600} bash -c "$cmd" > "$O/$label.txt" 2>&1; rc=$?
    if [ $rc -ne 0 ]; then tail -25 "$O/$label.txt"; exit $rc; fi
    if [ -n "$GREP" ]; then grep -E "$GREP" "$O/$label.txt" | cut -c1-220; else grep -v "amdgpu.ids" "$O/$label.txt" | tail -${LAB_TAIL:-40} | cut -c1-220; fi
  fi
done
