#!/bin/bash
# Round 5 lab: one library per MRG_X3Q_DBG value (timing-only variants of rowgemm_x3q_k: csrc/gemm_x3q.hpp) into tools/labso/.
#   tools/lab/q_parts_build.sh 1 2 4 8 16 32 20
set -e
cd "$(dirname "$0")/../../mr-gnas_amd/csrc"
mkdir -p ../../tools/labso/obj
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function"
for v in "$@"; do
  ( hipcc $FLAGS -DMRG_X3Q_DBG=$v -c linear.hip -o ../../tools/labso/obj/linear_q$v.o &
    hipcc $FLAGS -DMRG_X3Q_DBG=$v -c dense.hip -o ../../tools/labso/obj/dense_q$v.o &
    wait
    OTHERS=$(ls build/*.o | grep -v "build/linear.o\|build/dense.o")
    hipcc -shared -fPIC --offload-arch=gfx950 -o ../../tools/labso/libmrgnas_q$v.so ../../tools/labso/obj/linear_q$v.o ../../tools/labso/obj/dense_q$v.o $OTHERS ) &
  if (( $(jobs -r | wc -l) >= 4 )); then wait -n; fi
done
wait
ls -la ../../tools/labso/*.so
