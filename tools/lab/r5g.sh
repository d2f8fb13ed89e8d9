#!/bin/bash
# round 4, call 5g: the profile set at HEAD again (after the eight-tile block, the reroutes and the package split): kernel stats (both
# stream views), PMC traffic passes, SQ counters (default workload and c5_fixed_cell)
set -e
cd $GRAFT_REPO_ROOT
bash tools/profile_r4.sh r5prof 1234
bash tools/pmc_sq.sh r5sq
bash tools/pmc_sq.sh r5sq_c5 --workload c5_fixed_cell
ls gpurun_out/r5prof gpurun_out/r5sq gpurun_out/r5sq_c5
