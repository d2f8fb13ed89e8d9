O=$GRAFT_REPO_ROOT/gpurun_out/r3r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_plans_gpu.py tests/test_fullsize_gpu.py tests/test_compgcn_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?
tail -3 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3r/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "loss", d["loss"])
for k in ("mrg_span_gcs",):
    v=d["kernels"][k]; print(k, v["launches"], v["ms_total"], v["us_per_launch"])
print({k:(v["us_per_launch"], v["frac"], v.get("frac_compulsory")) for k,v in d.items() if k.startswith("north_star")})
PY
exit 0
