#!/usr/bin/env python3
"""Lab: the north-star kernel (mrg_span_gcs, compose = sub) on the FB15k-237 and C5 shapes for a span size given by MRG_SPAN and a
library build given by MRG_LIB_PATH (prefetch depth variants)."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from mr_gnas_amd import _lib, graph as G, synth
bench._lib = _lib
torch.cuda.set_device(0)
out = {"span": os.environ.get("MRG_SPAN", "96"), "lib": os.path.basename(os.environ.get("MRG_LIB_PATH", "default"))}
n, r, t = synth.SHAPES["fb15k237"]
g = G.build_search_graph(n, r, synth.synth_kg(n, r, t, 0), device="cuda")
out["fb"] = bench.north_star_kernel(g, 200)["us_per_launch"]
del g
if "--no-c5" not in sys.argv:
    out["c5"] = bench.north_star_c5(torch.device("cuda", 0))["us_per_launch"]
print(json.dumps(out))
