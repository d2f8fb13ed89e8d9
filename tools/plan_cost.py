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
    step.g = G.build_search_graph(len(node_id), R, gtri, device=dev)  # a "new" step graph (device build): every cached plan is gone
    step.src_in, _, _ = step.g.edges(form="all")
    step.edge_type = step.g.edata["e_type"]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t1))
print(f"{wl}: steady step {t_step * 1e3:.1f} ms; new graph: build {min(t[0] for t in ts) * 1e3:.1f} ms, first step on it "
      f"(all plans built lazily) {min(t[1] for t in ts) * 1e3:.1f} ms")

# the whole per-step data preparation on the device: sample + relabel + negatives + graph build
from mr_gnas_amd import sampler as SM, synth
ds = wl.split("_")[0]
Nn, Rr, Tt = synth.SHAPES[ds]
tri = torch.from_numpy(synth.synth_kg(Nn, Rr, Tt, 0)).to(dev)
size = {"30k": 30000, "300": 300}.get(wl.split("_")[2], 30000)
for _ in range(2):
    SM.generate_sampled_graph_and_labels(tri, size, 0.5, Rr, 10, Nn)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    SM.generate_sampled_graph_and_labels(tri, size, 0.5, Rr, 10, Nn)
torch.cuda.synchronize()
print(f"device sampler + negatives + graph build, graph_batch_size={size}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per step")
os.environ["MRG_TORCH_PLANS"] = "1"
t0 = time.perf_counter()
step.g = G.build_search_graph(len(node_id), R, gtri).to(dev)
step.src_in, _, _ = step.g.edges(form="all")
step.edge_type = step.g.edata["e_type"]
torch.cuda.synchronize()
t1 = time.perf_counter()
step()
torch.cuda.synchronize()
print(f"round-1 path (host numpy build + torch tensor-op plans): build {(t1 - t0) * 1e3:.1f} ms, first step {(time.perf_counter() - t1) * 1e3:.1f} ms")
