#!/bin/bash
# Round 5 lab: rowgemm_x3q_k with one part removed at a time (timing only), next to rowgemm_x3s_k in the same process.
mkdir -p gpurun_out/r5
for v in 0 1 2 4 8 16 32 20; do
  if [ $v = 0 ]; then unset MRG_LIB_PATH; else export MRG_LIB_PATH=$PWD/tools/labso/libmrgnas_q$v.so; fi
  echo "== MRG_X3Q_DBG=$v"
  timeout -k 10 120 python tools/rowgemm_ab.py --only linear,pair3 --rounds 3 --reps 20 2>&1 | grep "^linear\|^pair3" | cut -c1-260
done
