O=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_nets_gpu.py tests/test_configs_gpu.py -x -q -m gpu -k "cell_zero or nets or supernet or c2 or c3 or c4 or sharded" > $O/pytest.txt 2>&1; rc=$?
tail -5 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench.json 2> $O/bench.err || exit 3
MRG_CELL_ZERO_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench_unfused.json 2> $O/bench_unfused.err || exit 4
grep -h -o '"ms_per_step": [0-9.]*' $O/bench.json $O/bench_unfused.json
exit 0
