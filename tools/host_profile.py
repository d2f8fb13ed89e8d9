#!/usr/bin/env python3
"""Lab: where does the host time of one launch-bound search step go?  (cProfile over 20 steps of the 300-edge workload)"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

sys.argv = [sys.argv[0], "--workload", "fb15k237_supernet_300"] + sys.argv[1:]      # e.g. --workload fb15k237_supernet_30k --resample
args = bench.parse()
torch.cuda.set_device(0)
step = bench.Step(args, torch.device("cuda", 0), bench.build_step_inputs(args.workload, args.negative, args.seed))
for _ in range(5):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(45)
