set -x
mkdir -p gpurun_out/r3a
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 tools/labbin/gemm_x3_lab 272115 200 0 200 20 > gpurun_out/r3a/lab_272k.txt 2>&1
timeout -k 10 300 tools/labbin/gemm_x3_lab 558771 200 0 200 20 > gpurun_out/r3a/lab_558k.txt 2>&1
timeout -k 10 300 tools/labbin/gemm_x3_lab 272115 200 200 200 20 > gpurun_out/r3a/lab_272k_dual.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "row_order or fused_amean or fused_amax or split_core or grouped" > gpurun_out/r3a/pytest_ops.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/r3a/pytest_ops.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
echo "bench rc=$?"
tail -c 600 gpurun_out/r3a/pytest_ops.txt
