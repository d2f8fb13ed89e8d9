#!/bin/bash
# round 4, call r: the eight-tile single-block row GEMM (D = 256): bit identity and timing in the lab, GPU tests, the C5 bench line
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4r
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 300 tools/labbin/gemm_x3_lab_w8 2000000 256 0 256 3 > $O/lab_2m_256.txt 2>&1
grep -E "x3s8|x3s, 2|x3s accumulate, 2|float64" $O/lab_2m_256.txt
timeout -k 5 300 tools/labbin/gemm_x3_lab_w8 70001 256 0 240 3 > $O/lab_70k_240.txt 2>&1
grep -E "x3s8|x3s, 2|float64" $O/lab_70k_240.txt
python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py tests/test_host_cpu.py -x -q -m gpu -k "split_core or fused or c5 or amax or amean or bit_exact" > $O/pytest.txt 2>&1 || { tail -40 $O/pytest.txt; exit 1; }
tail -3 $O/pytest.txt
python bench.py --workload c5_fixed_cell --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_c5_fixed_cell.json 2> $O/bench_c5.err || { tail -30 $O/bench_c5.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r4r/bench_c5_fixed_cell.json')); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['loss'])"
