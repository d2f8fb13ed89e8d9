O=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $O
cd $GRAFT_REPO_ROOT
for ms in 4 1 2 4 1; do
  MRG_MIXED_STREAMS=$ms python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_ms$ms.json 2> $O/bench_ms$ms.err || exit 3
  echo "streams $ms: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_ms$ms.json)"
done
MRG_FORCE_SHARDED=1 python bench.py --workload fb15k237_supernet_30k --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_sharded_30k.json 2> $O/bench_sharded_30k.err || exit 4
echo "sharded 30k eager: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_sharded_30k.json)"
python bench.py --workload fb15k237_supernet_30k --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_30k.json 2> $O/bench_30k.err || exit 5
echo "plain 30k eager: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_30k.json)"
python bench.py --workload fb15k237_supernet_30k --hip-graph --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_30k_graph.json 2> $O/bench_30k_graph.err || exit 6
echo "plain 30k graph: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_30k_graph.json)"
MRG_FORCE_SHARDED=1 MRG_GRAPH_SHARDED=1 timeout -k 10 300 python bench.py --workload fb15k237_supernet_30k --hip-graph --steps 30 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_sharded_30k_graph.json 2> $O/bench_sharded_30k_graph.err
echo "sharded 30k graph rc=$?: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_sharded_30k_graph.json)"
python bench.py --workload wn18rr_supernet_full --steps 20 --warmup 5 --no-cpu-baseline --no-exact-f32-leg --no-c5 > $O/bench_wn18rr.json 2> $O/bench_wn18rr.err || exit 7
echo "wn18rr: $(grep -o '"ms_per_step": [0-9.]*, "higher' $O/bench_wn18rr.json)"
exit 0
