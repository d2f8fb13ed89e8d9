#!/bin/bash
# round 4, call m: the profile set at HEAD -- kernel stats (both stream views), PMC traffic passes, SQ counters (default workload and c5_fixed_cell)
set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_r4.sh r4prof 1234
bash tools/pmc_sq.sh r4sq
bash tools/pmc_sq.sh r4sq_c5 --workload c5_fixed_cell
ls gpurun_out/r4prof gpurun_out/r4sq gpurun_out/r4sq_c5
