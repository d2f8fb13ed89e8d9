#!/bin/bash
# round 4, call 5j (HEAD after the package split, the eight-tile block and the reroutes): the full GPU test suite (parity margins) and the bench lines of every single-GPU configuration at HEAD
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5j
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1 || { tail -60 $O/pytest_gpu.txt; exit 1; }
tail -3 $O/pytest_gpu.txt
cp gpurun_out/parity_margins.json $O/parity_margins.json
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
for w in fb15k237_fixed_d64 wn18rr_supernet_full fb15k237_supernet_30k fb15k237_supernet_300; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-c5 --no-cpu-baseline --no-caller-leg > $O/bench_$w.json 2> $O/bench_$w.err || { tail -30 $O/bench_$w.err; exit 1; }
done
for w in fb15k237_supernet_30k fb15k237_supernet_300; do
  python bench.py --workload $w --steps 20 --warmup 5 --no-c5 --no-cpu-baseline --no-caller-leg --hip-graph > $O/bench_${w}_hipgraph.json 2> $O/bench_${w}_hipgraph.err || { tail -30 $O/bench_${w}_hipgraph.err; exit 1; }
  python bench.py --workload $w --steps 20 --warmup 5 --no-c5 --no-cpu-baseline --no-caller-leg --resample > $O/bench_${w}_resample.json 2> $O/bench_${w}_resample.err || { tail -30 $O/bench_${w}_resample.err; exit 1; }
done
for c in sub mul ccorr; do
  python bench.py --workload compgcn_fb15k237 --comp-fn $c --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_compgcn_$c.json 2> $O/bench_compgcn_$c.err || { tail -30 $O/bench_compgcn_$c.err; exit 1; }
done
for r in 0 3 7; do
  python bench.py --rehearse-shard $r/8 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearse_${r}_replay.json 2> $O/rehearse_${r}_replay.err || { tail -30 $O/rehearse_${r}_replay.err; exit 1; }
done
python bench.py --workload c5_fixed_cell --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_fixed_cell.json 2> $O/bench_c5_fixed_cell.err || { tail -30 $O/bench_c5_fixed_cell.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5j/bench*.json")) + sorted(glob.glob("gpurun_out/r5j/rehearse*.json")):
    d=json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"]["launch"], d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"))
PY
