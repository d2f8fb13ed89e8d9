#!/bin/bash
# round 4, call i: sharded step on RCCL bound directly -- GPU tests (eager + replayed), FORCE_SHARDED world-1 bench (eager vs captured),
# and the timing-only rehearsal of one rank of eight
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4i
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_dist_gpu.py tests/test_nets_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
for r in 0 3 7; do
  python bench.py --rehearse-shard $r/8 --steps 20 --warmup 5 --no-cpu-baseline > $O/rehearse_${r}_replay.json 2> $O/rehearse_${r}_replay.err || { tail -30 $O/rehearse_${r}_replay.err; exit 1; }
done
python bench.py --rehearse-shard 3/8 --steps 20 --warmup 5 --no-cpu-baseline --no-shard-graph > $O/rehearse_3_eager.json 2> $O/rehearse_3_eager.err || { tail -30 $O/rehearse_3_eager.err; exit 1; }
MRG_FORCE_SHARDED=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/world1_direct.json 2> $O/world1_direct.err || { tail -30 $O/world1_direct.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r4i/*.json")):
    d=json.load(open(f))
    print(f.split("/")[-1], d.get("ms_per_step"), d["config"]["launch"], d["config"].get("rank_edges"), d["config"]["parallelism"][:80])
PY
