#!/usr/bin/env python3
"""Lab: does a tensor that one kernel has just written come back from the Infinity Cache (256 MB, memory side) when the next kernel
reads it?  out = a + b (mrg_sum_buffers: two reads, one write per element) repeated over the same three buffers, per-buffer size swept
across the cache size; then producer -> consumer: out = a + b followed by c = out + b, timed as a pair."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


print("MB per buffer | same three buffers again and again: GB/s | producer then consumer (5 buffer passes): GB/s")
for mb in (8, 16, 32, 64, 96, 128, 192, 256, 447, 894):
    n = mb * (1 << 20) // 4
    a, b = torch.randn(n, device=dev), torch.randn(n, device=dev)
    t1 = timeit(lambda: K.sum_buffers([a, b]))
    def pair():
        o = K.sum_buffers([a, b])
        return K.sum_buffers([o, b])
    t2 = timeit(pair)
    print(f"{mb:5d}   {3 * n * 4 / t1 / 1e9:8.0f}   {6 * n * 4 / t2 / 1e9:8.0f}")
    del a, b
