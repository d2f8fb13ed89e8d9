#!/usr/bin/env python3
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mr_gnas_amd import _lib, functional as K
lib = _lib.load()
dev = "cuda"
rows, Kd, N = 64, 64, 200
lib.mrg_gemm_set_q(1)
b = torch.zeros(N, device=dev)
# which x row lands in which output row
x = torch.zeros(rows, Kd, device=dev); x[:, 0] = torch.arange(rows, device=dev) + 1
W = torch.zeros(N, Kd, device=dev); W[:, 0] = 1
out = K.linear(x, W, b, None)
print("row test: out[:, 0] =", out[:20, 0].tolist())
print("row test: out[:, 17] =", out[:20, 17].tolist())
print("row test: out[:, 40] =", out[:20, 40].tolist())
# which W row (output column) lands in which output column
x = torch.zeros(rows, Kd, device=dev); x[:, 0] = 1
W = torch.zeros(N, Kd, device=dev); W[:, 0] = torch.arange(N, device=dev) + 1
out = K.linear(x, W, b, None)
print("col test: out[0, :40] =", out[0, :40].tolist())
print("col test: out[5, :40] =", out[5, :40].tolist())
# k positions
for kk in (0, 1, 5, 8, 17, 33, 63):
    x = torch.zeros(rows, Kd, device=dev); x[:, kk] = 1
    W = torch.zeros(N, Kd, device=dev); W[:, :] = torch.arange(Kd, device=dev) + 1
    out = K.linear(x, W, b, None)
    print("k test", kk, "->", out[0, 0].item(), out[3, 20].item(), out[9, 100].item())
