#!/usr/bin/env python3
"""One-off stress at the C5 shape (M = 11 M rows, D = 256: 2.8e9 elements per tensor, beyond 2^31): the fused MixedOp
epilogue, the K-way gradient sum and a dense filter, forward + backward, against float64 on a row sample."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K  # noqa: E402

dev = "cuda"
M, D = 11_000_000, 256
gen = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
rows = torch.cat((torch.arange(0, 2048, device=dev), torch.arange(M - 2048, M, device=dev), torch.randint(0, M, (2048,), device=dev, generator=gen)))

# K-way sum
xs = [rnd(M, D) for _ in range(3)]
tot = K.sum_buffers(xs)
ref = sum(x[rows].double() for x in xs)
print("sum_buffers max err", float((tot[rows].double() - ref).abs().max()))
del tot

# MixedOp epilogue forward/backward
bns = [torch.nn.BatchNorm1d(D).to(dev) for _ in range(3)]
w = torch.softmax(rnd(3), 0).requires_grad_(True)
ys = [x.requires_grad_(True) for x in xs]
out = K.mixed_epilogue(ys, bns, w, None, None)
refo = 0
for k in range(3):
    mean, var = ys[k].detach().double().mean(0), ys[k].detach().double().var(0, unbiased=False)
    z = (ys[k].detach()[rows].double() - mean) / torch.sqrt(var + bns[k].eps) * bns[k].weight.double() + bns[k].bias.double()
    refo = refo + w.detach().double()[k] * torch.relu(z)
print("mixed_epilogue fwd max err", float((out[rows].double() - refo).abs().max()))
g = rnd(M, D)
out.backward(g)
print("mixed_epilogue bwd finite", all(bool(torch.isfinite(y.grad[rows]).all()) for y in ys), "dw", w.grad.tolist())
del out, g, ys, xs
torch.cuda.empty_cache()

# dense filter (single segment) forward / backward
s, s_in = rnd(M, D).requires_grad_(True), rnd(M, D).requires_grad_(True)
W = (rnd(D, 2 * D) / (2 * D) ** 0.5).requires_grad_(True)
b = rnd(D).requires_grad_(True)
o = K.dense_filter_single(s, s_in, W, b)
cat = torch.cat((s.detach()[rows], s_in.detach()[rows]), 1).double()
refd = torch.sigmoid(cat @ W.detach().double().t() + b.detach().double()) * s.detach()[rows].double()
print("dense_filter fwd max err", float((o[rows].double() - refd).abs().max()))
o.backward(rnd(M, D))
print("dense_filter bwd finite", bool(torch.isfinite(W.grad).all()), bool(torch.isfinite(s.grad[rows]).all()), "peak GiB", round(torch.cuda.max_memory_allocated() / 2**30, 1))
