O=$GRAFT_REPO_ROOT/gpurun_out/r3p
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; rc=$?
tail -6 $O/pytest_gpu.txt
grep -q "Memory access fault" $O/pytest_gpu.txt && exit 9
[ $rc -ne 0 ] && exit $rc
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3p/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "loss", d["loss"])
for k in ("mrg_zero_stats_coef","mrg_zero_fwd","mrg_zero_bwd_reduce","mrg_zero_bwd_apply","mrg_span_gcs"):
    v=d["kernels"][k]; print(k, v["launches"], v["ms_total"], v["us_per_launch"])
print({k:(v["us_per_launch"], v["frac"], v.get("frac_compulsory")) for k,v in d.items() if k.startswith("north_star")})
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["ms_per_step"])
PY
exit 0
