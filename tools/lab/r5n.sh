#!/bin/bash
# round 4, call 5n: does the step gain when the row GEMM leaves half of every CU's registers to the HBM-bound kernels of other streams?
# (one GEMM workgroup per CU through unused LDS, candidates of a MixedOp on 1 / 2 / 4 streams)
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5n
mkdir -p $O
cd $GRAFT_REPO_ROOT
F="--steps 10 --warmup 3 --no-c5 --no-cpu-baseline --no-caller-leg --no-exact-f32-leg"
for ex in 0 40000; do for ms in 1 2 4; do
  MRG_X3S_LDS_EXTRA=$ex MRG_MIXED_STREAMS=$ms python bench.py $F > $O/b_${ex}_${ms}.json 2> $O/b_${ex}_${ms}.err || { tail -20 $O/b_${ex}_${ms}.err; exit 1; }
  python -c "
import json; d=json.load(open('$O/b_${ex}_${ms}.json')); print('lds_extra', $ex, 'mixed_streams', $ms, d['ms_per_step'], d['loss'])"
done; done
