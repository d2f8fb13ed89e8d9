O=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_dataprep_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; rc=$?
tail -15 $O/pytest.txt
grep -q "Memory access fault" $O/pytest.txt && exit 9
exit $rc
