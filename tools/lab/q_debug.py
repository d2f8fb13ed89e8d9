#!/usr/bin/env python3
"""Debug aid for rowgemm_x3q_k: where do its outputs differ from the float64 product?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mr_gnas_amd import _lib, functional as K
lib = _lib.load()
dev = "cuda"
torch.manual_seed(0)
for rows, Kd, N in ((64, 200, 200), (128, 64, 200), (70000, 200, 200)):
    x = torch.randn(rows, Kd, device=dev)
    W = torch.randn(N, Kd, device=dev) / Kd ** 0.5
    b = torch.zeros(N, device=dev)
    ref = (x.double() @ W.double().t()).float()
    for q in (0, 1):
        lib.mrg_gemm_set_q(q)
        out = K.linear(x, W, b, None)
        err = (out - ref).abs()
        print(f"rows {rows} K {Kd} N {N} q={q}: max err {float(err.max()):.3e}")
        if q == 1 and float(err.max()) > 1e-3:
            bad = err > 1e-3
            print("  bad share", float(bad.float().mean()))
            print("  bad by row%16:", [round(float(bad[r::16].float().mean()), 2) for r in range(16)])
            print("  bad by col//16:", [round(float(bad[:, c * 16:(c + 1) * 16].float().mean()), 2) for c in range((N + 15) // 16)])
            print("  bad by col%16:", [round(float(bad[:, c::16].float().mean()), 2) for c in range(16)])
            # is out a permutation of ref within a 16-row strip?
            o, r = out[:16], ref[:16]
            for rr in range(4):
                d = (o[rr:rr + 1, :32].unsqueeze(1) - r[:, :32].unsqueeze(0)).abs()     # [1,16,32]
                print("  out row", rr, "cols 0..7 match ref (row) at:", [int((r[:, c] - o[rr, c]).abs().argmin()) for c in range(8)],
                      "err", [round(float((r[:, c] - o[rr, c]).abs().min()), 4) for c in range(8)])
    lib.mrg_gemm_set_q(1)
