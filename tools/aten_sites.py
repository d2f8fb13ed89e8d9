#!/usr/bin/env python3
"""Lab: which Python lines of this repository issue the torch (aten) kernels of one step?
A TorchDispatchMode around one step with single-threaded autograd (the backward runs on the calling thread, so its ops are seen too);
every non-view aten op is attributed to the innermost frame inside the repository.  Usage: tools/aten_sites.py [bench.py arguments]"""
import collections
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

VIEWS = {"view", "_unsafe_view", "slice", "select", "t", "transpose", "expand", "as_strided", "detach", "alias", "unbind", "unsqueeze",
         "squeeze", "permute", "reshape", "narrow", "split", "split_with_sizes", "chunk", "unfold", "_reshape_alias", "lift_fresh",
         "empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "size", "stride", "is_same_size", "sym_size",
         "_local_scalar_dense", "item", "result_type", "can_cast", "record_stream", "set_", "resize_", "numel", "dim", "sym_numel",
         "sym_stride", "sym_storage_offset", "is_contiguous", "is_pinned", "_has_compatible_shallow_copy_type", "view_as_real"}


class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
        if name not in VIEWS:
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=40)):
                if fr.filename.startswith(ROOT) and "/tools/" not in fr.filename:
                    site = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            self.count[(name, site)] += 1
        return func(*args, **(kwargs or {}))


args = bench.parse()
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
if args.rehearse_shard:                               # one rank of a W-way sharded step on one GPU (bench.py --rehearse-shard)
    from mr_gnas_amd import cell_lp as CL, dist as MD, functional as K, rccl
    r_, w_ = (int(v) for v in args.rehearse_shard.split("/"))
    CL.MIXED_STREAMS = 1
    K.switches.SEGMENT_STREAMS = 1
    step = MD.ShardedStep(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed), r_, w_,
                          group=rccl.VirtualWorld(r_, w_, dev))
else:
    step = bench.Step(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(2):
    step()
torch.cuda.synchronize()
torch.autograd.set_multithreading_enabled(False)
with Sites() as s:
    step()
torch.cuda.synchronize()
by_site = collections.Counter()
for (name, site), n in s.count.items():
    by_site[site] += n
print(f"{sum(s.count.values())} non-view aten calls in one step")
for (name, site), n in s.count.most_common(70):
    print(f"{n:4d}  aten::{name:28s} {site}")
