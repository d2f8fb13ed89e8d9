O=$GRAFT_REPO_ROOT/gpurun_out/r3s
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_plans_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1 || { tail -5 $O/pytest.txt; exit 2; }
for snap in auto 0 3 6 12 24; do
  if [ $snap = auto ]; then unset MRG_SPAN_SNAP; else export MRG_SPAN_SNAP=$snap; fi
  python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-exact-f32-leg > $O/bench_snap$snap.json 2> $O/bench_snap$snap.err || exit 3
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3s/bench_snap$snap.json").read().strip().splitlines()[-1])
print("snap $snap:", "FB %.1f us frac %.3f | C5 %.1f us frac %.3f | step span_gcs %.3f ms | step %.2f ms" % (d["north_star_kernel"]["us_per_launch"], d["north_star_kernel"]["frac"], d["north_star_kernel_c5"]["us_per_launch"], d["north_star_kernel_c5"]["frac"], d["kernels"]["mrg_span_gcs"]["ms_total"], d["ms_per_step"]))
PY
done
exit 0
