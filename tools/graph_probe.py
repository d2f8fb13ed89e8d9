#!/usr/bin/env python3
"""Lab: can one whole search step (fwd + loss + bwd + clip + SGD) be captured in a HIP graph and replayed?
usage: python tools/graph_probe.py [workload] [steps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "fb15k237_supernet_300"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    sys.argv = [sys.argv[0], "--workload", wl]
    args = bench.parse()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    step = bench.Step(args, dev, bench.build_step_inputs(args.workload, args.negative, args.seed))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    import copy
    saved = (copy.deepcopy(step.model.state_dict()), copy.deepcopy(step.opt.state_dict()), torch.cuda.get_rng_state())
    t0 = time.perf_counter()
    losses = []
    for _ in range(steps):
        step()
        losses.append(step.last_loss.clone())
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / steps * 1e3
    print(f"eager {eager:.2f} ms/step losses {[round(float(l), 4) for l in losses]}", flush=True)
    step.model.load_state_dict(saved[0]); step.opt.load_state_dict(saved[1]); torch.cuda.set_rng_state(saved[2])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    step.model.load_state_dict(saved[0]); step.opt.load_state_dict(saved[1]); torch.cuda.set_rng_state(saved[2])
    t0 = time.perf_counter()
    losses = []
    for _ in range(steps):
        g.replay()
        losses.append(step.last_loss.clone())
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / steps * 1e3
    print(f"graph replay {rep:.2f} ms/step losses {[round(float(l), 4) for l in losses]}", flush=True)


if __name__ == "__main__":
    main()
