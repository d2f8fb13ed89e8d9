O=$GRAFT_REPO_ROOT/gpurun_out/r3t
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/labbin/gemm_x3_lab 272115 200 0 200 20 > $O/lab_272k.txt 2>&1 && \
timeout -k 10 120 tools/labbin/gemm_x3_lab 558771 200 0 200 20 > $O/lab_558k.txt 2>&1 && \
timeout -k 10 120 tools/labbin/gemm_x3_lab 272115 200 200 200 20 > $O/lab_272k_dual.txt 2>&1 && \
timeout -k 10 120 tools/labbin/gemm_x3_lab 70001 64 0 64 20 > $O/lab_70k_64.txt 2>&1
for f in $O/*.txt; do echo "== $f"; grep -h "x3s\|acc-order\|two waves" $f; done
