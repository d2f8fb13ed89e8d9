#!/usr/bin/env python3
"""Which operations of the static search step still synchronise with the host?  (torch.cuda.set_sync_debug_mode('warn') around one
eager step: every synchronising torch call warns with its stack.)  Usage: python tools/lab/static_sync_probe.py [workload]"""
import os, sys, warnings, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
sys.argv = [sys.argv[0], "--workload", sys.argv[1] if len(sys.argv) > 1 else "fb15k237_supernet_300", "--resample", "--static-step"]
args = bench.parse()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
from mr_gnas_amd import cell_lp as CL, functional as KF
CL.MIXED_STREAMS = 1
KF.switches.SEGMENT_STREAMS = 1
step = bench.Step(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(3):
    step()
torch.cuda.synchronize()
seen = {}
def showwarning(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" not in str(message).lower():
        return
    st = [f for f in traceback.extract_stack() if "static_sync_probe" not in f.filename and "warnings.py" not in f.filename]
    key = tuple((f.filename.replace(ROOT + "/", "").replace("/usr/local/lib/python3.10/dist-packages/", ""), f.lineno) for f in st[-5:])
    seen[key] = seen.get(key, 0) + 1
warnings.showwarning = showwarning
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
step()
torch.cuda.set_sync_debug_mode("default")
torch.cuda.synchronize()
print(f"{sum(seen.values())} synchronising calls in one static step, {len(seen)} distinct sites:")
for k, n in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(f"  x{n}  " + "  <-  ".join(f"{f}:{l}" for f, l in reversed(k)))
