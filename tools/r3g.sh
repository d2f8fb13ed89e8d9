O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
cd $GRAFT_REPO_ROOT
set -e
for m in 0 5 4 0 5; do
  MRG_GEMM_MODE=$m python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench_m$m.json 2> $O/bench_m$m.err
  grep -h -o '"ms_per_step": [0-9.]*' $O/bench_m$m.json
done
MRG_GEMM_MODE=5 timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py -x -q -m gpu > $O/pytest_mode5.txt 2>&1; tail -3 $O/pytest_mode5.txt
