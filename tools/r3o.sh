O=$GRAFT_REPO_ROOT/gpurun_out/r3o
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "dense_pair or nets or supernet or c2 or c3 or c4 or sharded or mixed or epilogue" > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench.json 2> $O/bench.err || exit 3
MRG_FOLD_IDENTITY=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_nofold.json 2> $O/bench_nofold.err || exit 4
python - <<'PY'
import json
for f in ("bench","bench_nofold"):
    d=json.loads(open(f"gpurun_out/r3o/{f}.json").read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["loss"], d["kernels"]["mrg_mix_bwd_apply"]["ms_total"], d["kernels"]["mrg_sum_buffers"]["ms_total"])
PY
exit 0
