#!/usr/bin/env python3
"""Lab: run one libmrgnas entry point in a tight loop for ~6 s (for tools/clock_probe.sh)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K, _lib
what = sys.argv[1] if len(sys.argv) > 1 else "gemm"
dev = "cuda"
rows, D = 272115, 200
x = torch.randn(rows, D, device=dev); W = torch.randn(D, D, device=dev) / 14; b = torch.zeros(D, device=dev)
if what == "exact":
    _lib.load().mrg_gemm_set_mode(1)
t0 = time.time()
while time.time() - t0 < 6:
    for _ in range(50):
        if what == "copy":
            K.compose("sub", x, x)
        else:
            K.linear(x, W, b, None)
    torch.cuda.synchronize()
