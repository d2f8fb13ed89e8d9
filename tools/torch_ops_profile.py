#!/usr/bin/env python3
"""Lab: which torch (non-library) device kernels are left in one step, and which Python line launches them?
torch.profiler with stacks over 3 steps of a bench workload; device time of every kernel that is not an mrg:: kernel, grouped by
the innermost frame of this repository."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

args = bench.parse()
from mr_gnas_amd import cell_lp as _CL  # noqa: E402
_CL.CALLER = args.caller                              # --caller reference: the literal formulation (lazy handles unless MRG_LAZY=0)
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if args.rehearse_shard:                               # one rank of a W-way sharded step on one GPU (timing only, bench.py --rehearse-shard)
    from mr_gnas_amd import cell_lp as CL, dist as MD, functional as K, rccl
    r_, w_ = (int(v) for v in args.rehearse_shard.split("/"))
    CL.MIXED_STREAMS = 1
    K.switches.SEGMENT_STREAMS = 1
    step = MD.ShardedStep(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed), r_, w_,
                          group=rccl.VirtualWorld(r_, w_, dev))
elif args.workload == "c5_fixed_cell":
    step = bench.FixedStep(args, dev, shape="synthetic10m", dim=256, init_dim=64, nbase=64)
elif args.workload == "fb15k237_fixed_d64":
    step = bench.FixedStep(args, dev)
else:
    step = bench.Step(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        step()
    torch.cuda.synchronize()
by_site = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
for ev in prof.events():
    if not ev.kernels:
        continue
    dev_us = sum(k.duration for k in ev.kernels if "mrg::" not in k.name and not k.name.startswith("mrg"))
    names = [k.name for k in ev.kernels if "mrg::" not in k.name]
    if not names:
        continue
    site = "?"
    for fr in ev.stack:
        if ROOT in fr and "/tools/" not in fr:
            site = fr.replace(ROOT + "/", "")
            break
    rec = by_site[(site, ev.name)]
    rec[0] += len(names)
    rec[1] += dev_us
    for n in names:
        rec[2][n.split("<")[0][:60]] += 1
tot = sum(v[1] for v in by_site.values())
print(f"non-library device time: {tot / N / 1e3:.3f} ms/step")
for (site, op), (n, us, names) in sorted(by_site.items(), key=lambda kv: -kv[1][1])[:60]:
    print(f"{us / N:8.1f} us/step  n/step={n / N:5.1f}  {op:28s} {site[:90]:90s} {dict(names.most_common(2))}")
