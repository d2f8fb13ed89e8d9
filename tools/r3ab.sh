#!/bin/bash
# same-box A/B of an environment switch: bench twice each, alternating.   usage: tools/r3ab.sh VAR=on_value VAR=off_value [kernel entry points to print]
A=$1; B=$2; shift 2
mkdir -p gpurun_out/r3ab
for rep in 1 2; do
  for v in "$A" "$B"; do
    env $v timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32-leg --no-c5 > gpurun_out/r3ab/b.json 2> gpurun_out/r3ab/b.err || { tail -5 gpurun_out/r3ab/b.err; exit 1; }
    python3 - "$v" "$rep" "$@" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/r3ab/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "rep", sys.argv[2], "ms/step", d["ms_per_step"], " ".join(f'{n.replace("mrg_", "")} {d["kernels"][n]["ms_total"]:.3f}' for n in sys.argv[3:] if n in d["kernels"]))
PY
  done
done
