#!/bin/bash
# round 4, call 5q: soak -- 300 steps of the headline step and of the resampled 30 000-edge / 300-edge steps in one process each
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r5q
mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/soak.py > $O/soak_full.txt 2>&1 || { tail -20 $O/soak_full.txt; exit 1; }
python tools/soak.py --workload fb15k237_supernet_30k --resample > $O/soak_30k_resample.txt 2>&1 || { tail -20 $O/soak_30k_resample.txt; exit 1; }
python tools/soak.py --workload fb15k237_supernet_300 --resample > $O/soak_300_resample.txt 2>&1 || { tail -20 $O/soak_300_resample.txt; exit 1; }
for f in soak_full soak_30k_resample soak_300_resample; do echo "== $f"; grep "^step" $O/$f.txt; done
