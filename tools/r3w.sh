O=$GRAFT_REPO_ROOT/gpurun_out/r3w
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; rc=$?
tail -3 $O/pytest_gpu.txt
grep -q "Memory access fault" $O/pytest_gpu.txt && exit 9
[ $rc -ne 0 ] && exit $rc
cp gpurun_out/parity_margins.json $O/
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1 || { cat $O/smoke.txt; exit 4; }
cat $O/smoke.txt
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 3
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3w/bench.json").read().strip().splitlines()[-1])
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", d["roofline"]["kernel"], d["roofline"]["frac"], "cpu", d["cpu_baseline"])
PY
