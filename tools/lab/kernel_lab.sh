#!/bin/bash
# Build-and-run of a stand-alone kernel lab source (tools/*_lab.hip, tools/lab/*.hip: phase stamps, part removal, bit
# identity of a candidate kernel against the shipped one, peak probes).  Build here (hipcc cross-compiles) or on the box.
#
#   tools/lab/kernel_lab.sh build SRC.hip [-DSWITCH=..]...        -> tools/labbin/<name>[_<switches>]   (git-ignored, travels with gpurun)
#   tools/lab/kernel_lab.sh run TAG BIN 'ARGS' ['ARGS' ...]       run the binary once per argument list, output under gpurun_out/TAG/
#   tools/lab/kernel_lab.sh libs NAME=-DSWITCH=V:a.hip,b.hip [...]  one libmrgnas per lab switch -> tools/labso/libmrgnas_NAME.so (MRG_LIB_PATH selects it)
#
# Examples: phase stamps of the shipped row GEMM        kernel_lab.sh build tools/x3s_trace_lab.hip; kernel_lab.sh run stamps x3s_trace_lab '558771 200 200' '558771 400 200'
#           rowgemm_x3q_k with one part removed at a time  kernel_lab.sh libs noepi=-DMRG_X3Q_DBG=1:linear.hip,dense.hip noa=-DMRG_X3Q_DBG=2:linear.hip,dense.hip
#                                                          then  tools/lab/ab.sh parts 'MRG_LIB_PATH=tools/labso/libmrgnas_noepi.so' ... -- python tools/rowgemm_ab.py
set -eu
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R"
MODE=$1; shift
case "$MODE" in
  build)
    src=$1; shift; name=$(basename "$src" .hip); suffix=$(echo "$*" | tr -cd 'A-Za-z0-9=' | tr '=' '_')
    mkdir -p tools/labbin
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I mr-gnas_amd/csrc -I include -I tools/lab "$@" "$src" -o "tools/labbin/$name${suffix:+_$suffix}"
    echo "tools/labbin/$name${suffix:+_$suffix}";;
  run)
    tag=$1; bin=$2; shift 2; mkdir -p "gpurun_out/$tag"
    for a in "$@"; do echo "== $bin $a"; timeout -k 5 ${LAB_TIMEOUT:-120} "tools/labbin/$bin" $a 2>&1 | tee -a "gpurun_out/$tag/$bin.txt"; done;;
  libs)
    # NAME=FLAG:src1.hip,src2.hip -- recompile the listed csrc sources with FLAG, link them with the shipped objects of the rest
    ( cd mr-gnas_amd/csrc && make -j8 ARCH=gfx950 > /dev/null )
    mkdir -p tools/labso/obj
    for spec in "$@"; do
      name=${spec%%=*}; rest=${spec#*=}; flag=${rest%%:*}; srcs=${rest#*:}
      objs=""; skip=""
      for s in ${srcs//,/ }; do
        b=$(basename "$s" .hip)
        hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $flag -c "mr-gnas_amd/csrc/$s" -o "tools/labso/obj/${b}_$name.o"
        objs="$objs tools/labso/obj/${b}_$name.o"; skip="$skip|build/$b.o"
      done
      others=$(ls mr-gnas_amd/csrc/build/*.o | grep -Ev "${skip#|}")
      hipcc -shared -fPIC --offload-arch=gfx950 -o "tools/labso/libmrgnas_$name.so" $objs $others
      echo "MRG_LIB_PATH=tools/labso/libmrgnas_$name.so"
    done;;
  *) echo "usage: $0 build|run|libs ..." >&2; exit 2;;
esac
