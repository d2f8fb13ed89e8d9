O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; rc=$?
tail -5 $O/pytest_gpu.txt
grep -q "Memory access fault" $O/pytest_gpu.txt && exit 9
[ $rc -ne 0 ] && exit $rc
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || exit 3
grep -h -o '"ms_per_step": [0-9.]*' $O/bench.json
cp gpurun_out/parity_margins.json $O/ 2>/dev/null
exit 0
