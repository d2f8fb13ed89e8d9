set -x
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
cd $GRAFT_REPO_ROOT
for sb in 512 1024 2048 256; do
  MRG_STREAM_BLOCKS=$sb python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench_sb$sb.json 2> $O/bench_sb$sb.err || echo "FAILED $sb"
done
MRG_STREAM_BLOCKS=512 MRG_GEMM_EPILOGUE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg > $O/bench_sb512_epi0.json 2> $O/bench_sb512_epi0.err
grep -h -o '"ms_per_step": [0-9.]*' $O/*.json
