#!/bin/bash
# sharded path with batched statistics collectives: the world-1 RCCL tests, then the rank-size rehearsal on / off
set -o pipefail
mkdir -p gpurun_out/r3z
timeout -k 10 900 python3 -m pytest tests/test_nets_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "shard" > gpurun_out/r3z/tests.txt 2>&1 || { tail -40 gpurun_out/r3z/tests.txt; exit 1; }
tail -3 gpurun_out/r3z/tests.txt
for b in 1 0; do
  MRG_BATCH_STATS=$b MRG_FORCE_SHARDED=1 timeout -k 10 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-exact-f32-leg > gpurun_out/r3z/bench_full_$b.json 2> gpurun_out/r3z/bench_full_$b.err || { tail -20 gpurun_out/r3z/bench_full_$b.err; exit 1; }
  MRG_BATCH_STATS=$b MRG_FORCE_SHARDED=1 timeout -k 10 300 python3 bench.py --workload fb15k237_supernet_30k --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > gpurun_out/r3z/bench_30k_$b.json 2> gpurun_out/r3z/bench_30k_$b.err || { tail -20 gpurun_out/r3z/bench_30k_$b.err; exit 1; }
done
python3 - <<'PY'
import json
for w in ("full", "30k"):
    for b in (1, 0):
        d = json.loads(open(f"gpurun_out/r3z/bench_{w}_{b}.json").read().strip().splitlines()[-1])
        print(w, "batch_stats", b, "ms/step", d["ms_per_step"], "loss", d.get("loss"), d["config"].get("parallelism"))
PY
