#!/usr/bin/env python3
"""Lab: many steps of one bench workload in one process -- allocated / reserved HBM and the step time at intervals (leaks, fragmentation,
drift).  Usage: tools/soak.py [bench.py arguments] (e.g. --workload fb15k237_supernet_30k --resample)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

args = bench.parse()
from mr_gnas_amd import cell_lp as _CL  # noqa: E402
_CL.CALLER = args.caller                              # --caller reference: the reference's literal formulation on lazy handles
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
step = bench.Step(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed))
N = max(args.steps, 300)
t0 = None
for i in range(N + 1):
    if i % 100 == 0:
        torch.cuda.synchronize()
        now = time.perf_counter()
        rate = "" if t0 is None else f"  {(now - t0) * 10:7.2f} ms/step over the last 100"
        print(f"step {i:4d}: allocated {torch.cuda.memory_allocated() / 2**20:9.1f} MiB  reserved {torch.cuda.memory_reserved() / 2**20:9.1f} MiB"
              f"  peak {torch.cuda.max_memory_allocated() / 2**20:9.1f} MiB{rate}  loss {float(step.last_loss) if step.last_loss is not None else float('nan'):.5f}", flush=True)
        t0 = time.perf_counter()
    step()
