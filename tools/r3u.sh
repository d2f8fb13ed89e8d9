O=$GRAFT_REPO_ROOT/gpurun_out/r3u
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split_core or fused or grouped or dense_pair" > $O/pytest.txt 2>&1 || { tail -5 $O/pytest.txt; exit 2; }
tail -1 $O/pytest.txt
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench$i.json 2> $O/bench$i.err || exit 3
python - <<PY
import json
d=json.loads(open("gpurun_out/r3u/bench$i.json").read().strip().splitlines()[-1])
f=d["kernel_families"]["row_gemm [rowgemm_x3s_k]"]
print("ms/step", d["ms_per_step"], "row_gemm", f["ms_total"], f["frac"])
PY
done
