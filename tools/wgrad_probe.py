#!/usr/bin/env python3
"""Lab: the split-core weight gradient through the C ABI, fragments split once per workgroup, the two waves of a SIMD in opposite phases (variant 1) against split per consuming wave (variant 0)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import _lib  # noqa: E402
from mr_gnas_amd._lib import call, ptr, stream_of  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda", 0)
for rows, K1, K2, Nout in ((558771, 200, 200, 200), (558771, 200, 0, 200), (558771, 190, 0, 200), (558771, 128, 0, 128), (558771, 100, 0, 100),
                           (70000, 200, 200, 200)):
    gy = torch.randn(rows, Nout, device=dev)
    x1 = torch.randn(rows, K1, device=dev)
    x2 = torch.randn(rows, K2, device=dev) if K2 else None
    gW, gb = torch.empty(Nout, K1 + K2, device=dev), torch.empty(Nout, device=dev)
    ws = torch.empty(max(16, int(lib.mrg_linear_bwd_weight_workspace_bytes(rows, K1 + K2, Nout))), dtype=torch.uint8, device=dev)
    line = f"rows {rows} K {K1}+{K2} Nout {Nout}:"
    for variant in (0, 1, 0, 1):
        lib.mrg_wgrad_set_variant(variant)
        go = lambda: call("mrg_linear_bwd_weight", (ptr(gy), ptr(x1), ptr(x2), ptr(gW), ptr(gb), ptr(ws), rows, K1, K2, Nout, stream_of(gW)))
        for _ in range(3):
            go()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            go()
        b.record()
        torch.cuda.synchronize()
        line += f"  v{variant} {a.elapsed_time(b) / 20 * 1e3:7.1f} us"
    print(line)
lib.mrg_wgrad_set_variant(1)
