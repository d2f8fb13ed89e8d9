O=$GRAFT_REPO_ROOT/gpurun_out/r3m
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py tests/test_nets_gpu.py -x -q -m gpu -k "c5_fixed_cell or c1 or fixed" > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --workload c5_fixed_cell --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_fixed_cell.json 2> $O/bench_c5.err || exit 3
timeout -k 10 900 python bench.py --workload fb15k237_fixed_d64 --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_fixed_d64.json 2> $O/bench_d64.err || exit 4
grep -h -o '"ms_per_step": [0-9.]*, "higher' $O/*.json
