#!/bin/bash
# SQ counters of one benchmark step (separate pass, kernel trace only): MFMA pipe busy cycles, wave cycles, waits, LDS bank conflicts.
# usage (through gpurun): tools/pmc_sq.sh TAG [bench.py arguments, e.g. --workload c5_fixed_cell]
#                          -> gpurun_out/TAG/pmc_sq/*counter_collection.csv, gpurun_out/TAG/sq_summary.txt
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/pmc_sq; mkdir -p $O/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exact-f32-leg --no-caller-leg --no-c5 "$@" > $O/pmc_sq.json 2> $O/pmc_sq.err || { tail -5 $O/pmc_sq.err; exit 1; }
find $O/pmc_sq -name "*kernel_trace.csv" -delete; find $O/pmc_sq -name "*agent_info.csv" -delete
python3 $R/tools/pmc_sq_summary.py $O/pmc_sq > $O/sq_summary.txt 2>&1
cat $O/sq_summary.txt
