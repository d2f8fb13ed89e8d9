#!/bin/bash
# round 4, call f: (1) the row GEMM's epilogue alone (store pattern at the kernel's grid), both accumulator layouts; (2) MFMA stream beside a partner wave
set -e
O=$GRAFT_REPO_ROOT/gpurun_out/r4f
mkdir -p $O
cd $GRAFT_REPO_ROOT
for m in 0 2 0 2; do timeout -k 5 60 tools/labbin/x3s_dbg_256 558771 200 200 $m >> $O/epilogue_only.txt 2>&1; done
for m in 0 2; do timeout -k 5 60 tools/labbin/x3s_dbg_1 558771 200 200 $m >> $O/epilogue_only.txt 2>&1; done
for m in 0 2; do timeout -k 5 60 tools/labbin/x3s_dbg_0 558771 200 200 $m >> $O/epilogue_only.txt 2>&1; done
timeout -k 5 120 tools/labbin/mfma_partner 2000 > $O/partner.txt 2>&1
cat $O/epilogue_only.txt $O/partner.txt
