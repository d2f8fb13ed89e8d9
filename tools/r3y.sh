#!/bin/bash
# split-once weight gradient: parity tests, then the bench with the variant on / off
set -o pipefail
mkdir -p gpurun_out/r3y
timeout -k 10 600 python3 -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "weight_gradient or linear" > gpurun_out/r3y/tests.txt 2>&1 || { tail -40 gpurun_out/r3y/tests.txt; exit 1; }
tail -3 gpurun_out/r3y/tests.txt
timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32-leg > gpurun_out/r3y/bench_on.json 2> gpurun_out/r3y/bench_on.err || { tail -20 gpurun_out/r3y/bench_on.err; exit 1; }
MRG_WGRAD_VARIANT=0 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32-leg > gpurun_out/r3y/bench_off.json 2> gpurun_out/r3y/bench_off.err || { tail -20 gpurun_out/r3y/bench_off.err; exit 1; }
python3 - <<'PY'
import json
r = {}
for t in ("on", "off"):
    d = r[t] = json.loads(open(f"gpurun_out/r3y/bench_{t}.json").read().strip().splitlines()[-1])
    print(t, "ms/step", d["ms_per_step"], "value", d["value"], "loss", d.get("loss"))
ko, kf = r["on"]["kernels"], r["off"]["kernels"]
for n in sorted(set(ko) | set(kf)):
    ta, tb = ko.get(n, {}).get("ms_total", 0.0), kf.get(n, {}).get("ms_total", 0.0)
    if abs(ta - tb) > 0.03:
        print(f"{n:36s} on {ta:7.3f} off {tb:7.3f}  d {ta - tb:+.3f}")
PY
