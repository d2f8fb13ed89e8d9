#!/usr/bin/env python3
"""Lab: cost of building a step graph and every index plan a search step touches (the reference samples a new
step graph per step, search/mr_lp_search.py:187-245), against the time of the step itself."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_supernet_30k"
sys.argv = [sys.argv[0], "--workload", wl]
args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
inputs = bench.build_step_inputs(args.workload, args.negative, args.seed)
step = bench.Step(args, dev, inputs)
for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    step()
torch.cuda.synchronize()
t_step = (time.perf_counter() - t0) / 5
from mr_gnas_amd import graph as G
N, R, node_id, gtri, samples, labels = inputs
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.g = G.build_search_graph(len(node_id), R, gtri).to(dev)      # a "new" step graph: every cached plan is gone
    step.src_in, _, _ = step.g.edges(form="all")
    step.edge_type = step.g.edata["e_type"]
    step.model._gather_cache = {}
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
print(f"{wl}: steady step {t_step * 1e3:.1f} ms; new graph: build {min(t[0] for t in ts) * 1e3:.1f} ms, first step on it "
      f"(all plans built lazily) {min(t[1] for t in ts) * 1e3:.1f} ms")
