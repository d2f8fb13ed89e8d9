#!/usr/bin/env python3
"""Lab: mrg_compose_bwd (mult / sub, both gradients) at the headline and the C5 row counts: GB/s of its algorithmic bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K  # noqa: E402

dev = torch.device("cuda", 0)
for rows, D in ((558771, 200), (5500000, 256), (11000000, 256)):
    s = torch.randn(rows, D, device=dev, requires_grad=True)
    hr = torch.randn(rows, D, device=dev, requires_grad=True)
    g = torch.randn(rows, D, device=dev)
    for op, nt in (("mult", 5), ("sub", 3)):
        out = K.compose(op, s, hr)
        def go():
            s.grad = hr.grad = None
            out.backward(g, retain_graph=True)
        for _ in range(3):
            go()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            go()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 10
        print(f"rows {rows:9d} D {D} compose_bwd {op:4s}: {ms * 1e3:9.1f} us  {nt * rows * D * 4 / ms / 1e6:7.0f} GB/s")
    del s, hr, g, out
