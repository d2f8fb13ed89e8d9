O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c5_fixed_cell" > $O/pytest.txt 2>&1; rc=$?
tail -15 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python bench.py --workload c5_fixed_cell --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_fixed_cell.json 2> $O/bench_c5.err; rc=$?
tail -3 $O/bench_c5.err; head -c 1500 $O/bench_c5_fixed_cell.json
exit $rc
