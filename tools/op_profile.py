#!/usr/bin/env python3
"""Lab: which torch operators (outside libmrgnas) launch the small kernels of one step?  torch.profiler, one step."""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_supernet_30k"
sys.argv = [sys.argv[0], "--workload", wl]
args = bench.parse()
torch.cuda.set_device(0)
step = bench.Step(args, torch.device("cuda", 0), bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="count", row_limit=45, max_name_column_width=60))
