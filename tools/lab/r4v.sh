#!/bin/bash
# round 4, call v: the [B, N] scorer input gradient on the split-over-rows kernel: tests, C5 probe, fixed d64 probe
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4v
mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_ops_gpu.py tests/test_dataprep_gpu.py -x -q -m gpu -k "linear or score" > $O/pytest.txt 2>&1 || { tail -30 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python tools/c5_probe.py > $O/c5_probe.txt 2>&1 || { tail -20 $O/c5_probe.txt; exit 1; }
cat $O/c5_probe.txt
python tools/c5_probe.py fixed64 > $O/fixed64_probe.txt 2>&1 || { tail -20 $O/fixed64_probe.txt; exit 1; }
cat $O/fixed64_probe.txt
