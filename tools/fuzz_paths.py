#!/usr/bin/env python3
"""Lab: randomized comparison of the launch-count optimisations against their per-segment / two-launch forms.
Random shapes (D multiple of 4 above 48, random edge counts and direction splits incl. empty segments); every output and
gradient must be bit-identical (a_mean: equal within summation-order tolerance; a_max's product-free input gradient: to rounding).  usage: python tools/fuzz_paths.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import functional as K, graph as G, operations_lp as O  # noqa: E402

DEV = "cuda"
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)


class SplitGraph(G.RelGraph):
    """RelGraph with an arbitrary direction split (b0, b1 = E): exercises ragged / empty direction segments."""
    def bounds(self):
        return self._b0, self.num_edges()


def run(op, g, a0, b0, gout, tied):
    a = a0.clone().requires_grad_(True)
    b = a if tied else b0.clone().requires_grad_(True)
    op.zero_grad()
    out = op(g, a, b)
    out.backward(gout)
    return [out.detach(), a.grad] + ([] if tied else [b.grad]) + [p.grad.clone() for p in op.parameters() if p.grad is not None]


bad = 0
for c in range(cases):
    D = int(rng.choice([52, 64, 100, 128, 200, 224, 256]))
    N = int(rng.integers(5, 3000))
    E = int(rng.choice([0, 1, 7, int(rng.integers(10, 5000)), int(rng.integers(5000, 200000))]))
    b0 = int(rng.choice([0, E, E // 2, int(rng.integers(0, E + 1))]))
    gen = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    src, dst = torch.randint(0, N, (E,), generator=gen), torch.randint(0, N, (E,), generator=gen)
    g = SplitGraph(N, src.numpy(), dst.numpy(), torch.randint(0, 6, (E,), generator=gen).numpy(),
                   (torch.rand(E, generator=gen) + 0.1).numpy().astype(np.float32), device=DEV)
    g._b0 = b0
    for kind in ("f_dense_comp", "f_comp", "a_max", "a_mean"):
        tied = bool(rng.integers(2)) and not kind.startswith("a_")
        op = O.MIXED_OPS[kind]({"feature_dim": D}).to(DEV)
        a0 = torch.randn(E + N, D, generator=gen).to(DEV)
        b0_ = torch.randn(E + N, D, generator=gen).to(DEV)
        gout = torch.randn(N if kind.startswith("a_") else E + N, D, generator=gen).to(DEV)
        res = {}
        keep = (K.switches.GROUPED_SEGMENTS, K.switches.FUSED_AMAX, K.switches.FUSED_AMAX_MIN_ROWS, K.switches.FUSED_AMEAN, K.switches.SPARSE_AMAX_BWD)
        try:
            K.switches.FUSED_AMAX_MIN_ROWS = 0
            # the bit comparison is between forms that share the dense input-gradient product; a_max's product-free input gradient
            # (round 5: mrg_segmax_bwd_input, exact f32 sums over the columns an edge won) is compared with it to rounding below
            K.switches.SPARSE_AMAX_BWD = False
            for fast in (True, False):
                K.switches.GROUPED_SEGMENTS = K.switches.FUSED_AMAX = K.switches.FUSED_AMEAN = fast
                res[fast] = run(op, g, a0, b0_, gout, tied)
            if kind == "a_max":
                K.switches.GROUPED_SEGMENTS = K.switches.FUSED_AMAX = K.switches.FUSED_AMEAN = True
                K.switches.SPARSE_AMAX_BWD = True
                res["sparse"] = run(op, g, a0, b0_, gout, tied)
        finally:
            (K.switches.GROUPED_SEGMENTS, K.switches.FUSED_AMAX, K.switches.FUSED_AMAX_MIN_ROWS, K.switches.FUSED_AMEAN,
             K.switches.SPARSE_AMAX_BWD) = keep
        def same(x, y):
            if x is None or y is None:
                return x is None and y is None
            if kind == "a_mean":                         # the fused form adds in another (fixed) order: tolerance, not bits
                return bool(((x - y).abs() <= 2e-5 * max(1.0, float(y.abs().max())) + 1e-6).all())
            return torch.equal(x, y)
        ok = len(res[True]) == len(res[False]) and all(same(x, y) for x, y in zip(res[True], res[False]))
        if ok and "sparse" in res:                         # same output, message gradient (hence weight / bias gradients) bit for bit; gx to rounding
            sp, de = res["sparse"], res[True]
            ok = len(sp) == len(de) and all(((x is None and y is None) or (x is not None and y is not None and torch.equal(x, y)))
                                            for i, (x, y) in enumerate(zip(sp, de)) if i != 1)
            ok = ok and bool(((sp[1] - de[1]).abs() <= 2e-5 * max(1.0, float(de[1].abs().max())) + 1e-6).all())
        if not ok:
            bad += 1
            print(f"MISMATCH case {c}: {kind} tied={tied} N={N} E={E} b0={b0} D={D}")
    if c % 10 == 9:
        print(f"{c + 1} cases done, {bad} mismatches", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
