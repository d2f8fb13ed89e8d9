O=$GRAFT_REPO_ROOT/gpurun_out/r3q
mkdir -p $O
cd $GRAFT_REPO_ROOT
MRG_TRACE=1 timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -m gpu -k "c2_fb15k237" > $O/pytest.txt 2> $O/trace.txt; rc=$?
tail -3 $O/pytest.txt; grep -v "^\[mrg\]" $O/trace.txt | head -20; grep "^\[mrg\]" $O/trace.txt | tail -4 | cut -c1-400
exit 0
