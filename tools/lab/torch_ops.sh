#!/bin/bash
# What is left to torch in a step: the aten kernels of one bench configuration by op (tools/torch_ops_profile.py), and
# optionally the Python lines that issue them (tools/aten_sites.py) and the host-side cProfile of the step (tools/host_profile.py).
#
#   tools/lab/torch_ops.sh TAG [ops|sites|host|all] [flags of the three tools: --workload .. --caller .. --resample ..]
#
# Examples: headline step                 tools/lab/torch_ops.sh ops_head all
#           the unchanged reference caller tools/lab/torch_ops.sh ops_ref ops --caller reference
#           one rank of the 8-way shard    tools/lab/torch_ops.sh ops_shard all --rehearse-shard 0/8
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
TAG=$1; WHAT=${2:-ops}; shift; [ $# -gt 0 ] && shift
O=gpurun_out/$TAG; mkdir -p "$O"
run() { timeout -k 10 ${LAB_TIMEOUT:-600} python "$1" "${@:3}" > "$O/$2.txt" 2>&1 || { tail -30 "$O/$2.txt"; exit 1; }; grep -v "amdgpu.ids" "$O/$2.txt" | head -${LAB_HEAD:-70}; }
case "$WHAT" in ops|all) run tools/torch_ops_profile.py torch_ops "$@";; esac
case "$WHAT" in sites|all) run tools/aten_sites.py aten_sites "$@";; esac
case "$WHAT" in host|all) run tools/host_profile.py host_profile "$@";; esac
