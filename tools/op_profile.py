#!/usr/bin/env python3
"""Lab: which torch operators / autograd nodes (outside libmrgnas) run in one step, by count and host time.
usage: python tools/op_profile.py [workload] [filter-substring]"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_supernet_30k"
flt = sys.argv[2] if len(sys.argv) > 2 else ""
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
rows = [e for e in prof.key_averages() if flt in e.key]
rows.sort(key=lambda e: -e.count)
for e in rows[:60]:
    print(f"{e.count:6d}  cpu {e.cpu_time_total / 1e3:8.3f} ms  self {e.self_cpu_time_total / 1e3:8.3f} ms   {e.key[:90]}")
