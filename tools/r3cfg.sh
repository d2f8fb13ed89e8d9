#!/bin/bash
# the other single-GPU configurations at HEAD -> gpurun_out/r3cfg/*.json (copied to profiles/r3_bench_*.json)
set -o pipefail
O=gpurun_out/r3cfg; mkdir -p $O
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" --no-cpu-baseline --no-exact-f32-leg > $O/$name.json 2> $O/$name.err || { tail -5 $O/$name.err; exit 1; }; python3 -c "
import json,sys; d=json.loads(open('$O/$name.json').read().strip().splitlines()[-1]); print('$name', d['ms_per_step'], d['value'], d['unit'])"; }
run fixed_d64 --workload fb15k237_fixed_d64 --steps 50 --warmup 10 || exit 1
run wn18rr --workload wn18rr_supernet_full --steps 20 --warmup 5 || exit 1
run 30k --workload fb15k237_supernet_30k --steps 40 --warmup 10 || exit 1
run 30k_graph --workload fb15k237_supernet_30k --steps 40 --warmup 10 --hip-graph || exit 1
run c5_fixed_cell --workload c5_fixed_cell --steps 5 --warmup 2 || exit 1
