#!/usr/bin/env python3
"""Lab: which Python lines issue the small ATen kernels (fill / copy / add / mul ...) of a launch-bound step?
torch.profiler with stacks over 5 steps of a bench workload; prints, per ATen op, the innermost repo frames and counts per step."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0], "--workload", "fb15k237_supernet_30k"] + sys.argv[1:]
args = bench.parse()
torch.cuda.set_device(0)
step = bench.Step(args, torch.device("cuda", 0), bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
want = ("aten::zero_", "aten::fill_", "aten::copy_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::zeros", "aten::contiguous",
        "aten::sum", "aten::cat", "aten::_to_copy", "aten::mul_")
agg = collections.Counter()
for ev in prof.events():
    if ev.name in want:
        frames = [f for f in (ev.stack or []) if "/repo/" in f and "aten_sources" not in f]
        where = frames[0].split("/repo/")[-1] if frames else "(autograd engine / torch internals)"
        agg[(ev.name, where)] += 1
for (name, where), c in agg.most_common(45):
    print(f"{c / N:6.1f}/step  {name:18s} {where}")
