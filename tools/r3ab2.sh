#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3ab
timeout -k 10 900 python3 -m pytest tests/test_nets_gpu.py tests/test_ops_gpu.py -x -q -m gpu -k "supernet or fan or mixed or agg or a_sum or golden" > gpurun_out/r3ab/tests.txt 2>&1 || { tail -40 gpurun_out/r3ab/tests.txt; exit 1; }
tail -2 gpurun_out/r3ab/tests.txt
bash tools/r3ab.sh MRG_LAZY_ASUM=1 MRG_LAZY_ASUM=0 mrg_sum_buffers mrg_sum_rows_gather mrg_seg_reduce_bwd
