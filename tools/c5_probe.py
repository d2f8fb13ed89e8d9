#!/usr/bin/env python3
"""Lab: per-launch times of the row-GEMM entry points inside one c5_fixed_cell step, eight-tile block on / off."""
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from mr_gnas_amd import _lib  # noqa: E402

args = types.SimpleNamespace(seed=0, dim=256, workload="c5_fixed_cell", negative=10)
SMALL = len(sys.argv) > 1 and sys.argv[1] == "fixed64"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lib = _lib.load()
step = bench.FixedStep(args, dev) if SMALL else bench.FixedStep(args, dev, shape="synthetic10m", dim=256, init_dim=64, nbase=64)
for _ in range(2):
    step()
torch.cuda.synchronize()
names = ["mrg_linear_bwd_input", "mrg_linear_fwd", "mrg_linear_relu_segmax_fwd", "mrg_linear_bwd_weight"]
for on in (0, 1, 0, 1):
    lib.mrg_gemm_set_wide8(on)
    step()
    torch.cuda.synchronize()
    _lib.meter.start(names)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    step()
    t1.record()
    torch.cuda.synchronize()
    recs = {k: [(a.elapsed_time(b), fl) for a, b, _, fl in v] for k, v in _lib.meter.records.items()}
    _lib.meter.stop()
    print(f"wide8={on}: step {t0.elapsed_time(t1):.1f} ms")
    for k, v in recs.items():
        print("   ", k, " ".join(f"{ms:.2f}ms/{fl / 1e12:.2f}TF" for ms, fl in v))
