#!/usr/bin/env python3
"""Lab: the MixedOp epilogue kernels with a stored f_dense_comp candidate vs the gate-only (recomputed) one, at the FB15k-237 first-stage
shape (rows = E + N, D = 200, K = 5: f_zero, f_identity, f_dense_comp, f_sparse_comp, f_comp).  Per-kernel HIP-event times."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mr_gnas_amd
from mr_gnas_amd import _lib
from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of

rows, D, K_ = 544230 + 14541, 200, 5
edge = 544230
dev = "cuda"
torch.manual_seed(0)
s = torch.randn(rows, D, device=dev)
gate = torch.rand(rows, D, device=dev)
norm = torch.rand(edge, device=dev)
yd = torch.empty_like(s)
yd[:edge] = gate[:edge] * s[:edge] * ((1.0 / 3.0) * norm)[:, None]
yd[edge:] = gate[edge:] * s[edge:] * (1.0 / 3.0)
y3, y4 = torch.randn(rows, D, device=dev), torch.randn(rows, D, device=dev)
g = torch.randn(rows, D, device=dev)
gam = [torch.rand(D, device=dev) + 0.5 for _ in range(K_)]
bet = [torch.randn(D, device=dev) * 0.1 for _ in range(K_)]
w = torch.softmax(torch.randn(K_, device=dev), 0)
lib = _lib.load()
ws = torch.empty(int(lib.mrg_mix_workspace_bytes(K_, D)), dtype=torch.uint8, device=dev)
st = stream_of(s)
cvec = torch.cat([norm * (1.0 / 3.0), torch.full((rows - edge,), 1.0 / 3.0, device=dev)])
spec = dict(k=2, s=s, c=cvec)
res = {}
for tag, ys, gspec in (("stored", [None, s, yd, y3, y4], None), ("gated", [None, s, gate, y3, y4], spec), ("gated-nopair", [None, y3.clone(), gate, y3, y4], spec)):
    ypa = ptr_array(ys)
    coef = torch.empty(K_, 4, D, device=dev)
    out = torch.empty(rows, D, device=dev)
    red = torch.empty(K_, 3, D, device=dev)
    coef2 = torch.zeros(K_, 2, D, device=dev)
    gys = [None] + [torch.empty(rows, D, device=dev) for _ in range(4)]
    gb = lambda: _lib.gated_branch(gspec)
    steps = {
        "stats": lambda: call("mrg_mix_stats_coef", (ypa, ptr_array(gam), ptr_array(bet), None, None, K_, rows, float(rows), D, 1e-5, 0.1, ptr(coef), ptr(ws), gb(), st)),
        "fwd": lambda: call("mrg_mix_fwd", (ypa, K_, ptr(coef), ptr(w), None, ptr(out), rows, D, gb(), st)),
        "reduce": lambda: call("mrg_mix_bwd_reduce", (ptr(g), ypa, K_, ptr(coef), ptr(w), ptr(red), ptr(ws), rows, D, gb(), st)),
        "apply": lambda: call("mrg_mix_bwd_apply", (ptr(g), ypa, ptr_array(gys), K_, ptr(coef), ptr(coef2), ptr(w), None, None, None, None, None, None, None, None, None, None,
                                                    rows, D, gb(), st)),
    }
    for name, fn in steps.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            fn()
        b.record()
        torch.cuda.synchronize()
        res[(tag, name)] = a.elapsed_time(b) / 20 * 1e3
    res[(tag, "out")] = out.clone()
    res[(tag, "coef")] = coef.clone()
for name in ("stats", "fwd", "reduce", "apply"):
    print(f"{name:8s} " + "  ".join(f"{tag} {res[(tag, name)]:8.1f} us" for tag in ("stored", "gated", "gated-nopair")))
print("bit-identical out:", torch.equal(res[("stored", "out")], res[("gated", "out")]), " coef:", torch.equal(res[("stored", "coef")], res[("gated", "coef")]))
