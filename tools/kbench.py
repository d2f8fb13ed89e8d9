#!/usr/bin/env python3
"""Per-kernel micro-benchmark of libmrgnas_hip.so at benchmark shapes (HIP events, median of
repeats).  Usage on the GPU box:  python tools/kbench.py [--shape fb|c5] [--only name,...]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import _lib, functional as K, graph as G, synth  # noqa: E402

HBM, MFMA = 8000.0, 157.3


def timeit(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="fb")
    ap.add_argument("--dim", type=int, default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    dev = "cuda"
    if args.shape == "fb":
        N, R, T = synth.SHAPES["fb15k237"]; D = args.dim or 200
    elif args.shape == "wn":
        N, R, T = synth.SHAPES["wn18rr"]; D = args.dim or 200
    else:
        N, R, T = synth.SHAPES["synthetic10m"]; D = args.dim or 256
    tri = synth.synth_kg(N, R, T, 0)
    g = G.build_search_graph(N, R, tri).to(dev)
    E = g.num_edges(); M = E + N
    src, dst, _ = g.edges(form="all")
    only = set(filter(None, args.only.split(",")))
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
    x, x_in, gM = rnd(M, D), rnd(M, D), rnd(M, D)
    gN = rnd(N, D)
    res = {}

    def rec(name, ms, nbytes=0, flops=0):
        med, mn = ms
        if flops:
            ach = flops / (med / 1e3) / 1e12; res[name] = dict(ms=round(med, 4), min_ms=round(mn, 4), tflops=round(ach, 2), frac=round(ach / MFMA, 4))
        else:
            ach = nbytes / (med / 1e3) / 1e9; res[name] = dict(ms=round(med, 4), min_ms=round(mn, 4), gbs=round(ach, 1), frac=round(ach / HBM, 4))
        print(name, res[name], flush=True)

    def want(n):
        return not only or n in only

    with torch.no_grad():
        if want("compose"):
            rec("compose_fwd", timeit(lambda: K._Compose.apply(1, x, x_in), args.reps), 12 * D * M)
        if want("gather"):
            ent = rnd(N, D); idx = torch.cat((src, torch.arange(N, device=dev))).int()
            rec("gather_fwd", timeit(lambda: K.gather_rows(ent, idx), args.reps), M * (8 * D + 4))
        if want("gate"):
            W = [rnd(D, 2 * D) * 0.05 for _ in range(3)]; b = [rnd(D) * 0.1 for _ in range(3)]; a = [rnd(1, D) * 0.1 for _ in range(3)]
            p = [t for i in range(3) for t in (W[i], b[i], a[i])]
            norm = g.norm_flat()
            rec("gate_fwd", timeit(lambda: K._Gate.forward(_Ctx(), x, x_in, norm, E // 2, E, 1 / 3, *p), args.reps), 12 * D * M + 4 * E)
            rec("gate_fwd_tied", timeit(lambda: K._Gate.forward(_Ctx(), x, x, norm, E // 2, E, 1 / 3, *p), args.reps), 8 * D * M + 4 * E)
    if want("gate"):
        xs = x.clone().requires_grad_(True); xi_ = x_in.clone().requires_grad_(True)
        out = K.gate_comp(xs, xi_, g.norm_flat(), E // 2, E, *p)
        rec("gate_bwd(+param)", timeit(lambda: torch.autograd.grad(out, (xs, xi_), gM, retain_graph=True), args.reps), 20 * D * M + 4 * E)
        out_t = K.gate_comp(xs, xs, g.norm_flat(), E // 2, E, *p)
        rec("gate_bwd_tied(+param)", timeit(lambda: torch.autograd.grad(out_t, xs, gM, retain_graph=True), args.reps), 12 * D * M + 4 * E)
    if want("seg"):
        for kind in ("sum", "max"):
            with torch.no_grad():
                rec(f"seg_{kind}_fwd", timeit(lambda: K._seg_fwd(K.REDUCE[kind], x, x[E:], g.plan(), N, D), args.reps),
                    4 * D * E + 4 * E + 4 * D * N * (2 + (kind == "max")))
            xs = x.clone().requires_grad_(True)
            out = K.aggregate_rows(kind, xs, g)
            rec(f"seg_{kind}_bwd", timeit(lambda: torch.autograd.grad(out, xs, gN, retain_graph=True), args.reps),
                4 * D * M + 4 * E + 4 * D * N * (1 + (kind == "max")))
    if want("gcs"):
        # the CompGCN aggregation: segments = (dst, direction)
        Rp = 2 * R + 1
        ent, rel = rnd(N, D), rnd(Rp, D)
        direction = (torch.arange(E, device=dev) < E // 2).long()
        cp = K.ComposePlan(src, g.edata["e_type"], dst * 2 + direction, g.norm_flat(), N, Rp, 2 * N)
        nb = E * (8 + 4 * D) + 4 * (2 * N + 1) + 4 * D * (Rp + 2 * N)
        with torch.no_grad():
            for kind in ("sub", "mul"):
                rec(f"chunk_gcs_{kind}", timeit(lambda: K.fused_gcs(kind, ent, cp.xi, rel, cp.yi, cp.scal, cp.by_seg, 2 * N), args.reps), nb)
                rec(f"span_gcs_{kind}", timeit(lambda: K.span_gcs(kind, ent, rel, cp.m_fwd, cp.sp_seg), args.reps), nb)
            G2 = rnd(2 * N, D)
            rec("span_gcs_bwd_node(copy)", timeit(lambda: K.span_gcs("copy", G2, None, cp.m_bx_unit(), cp.sp_x), args.reps), E * (8 + 4 * D) + 4 * D * N)
            rec("span_gcs_bwd_rel(negs)", timeit(lambda: K.span_gcs("negs", G2, None, cp.m_by_g, cp.sp_y), args.reps), E * (8 + 4 * D) + 4 * D * Rp)
            if args.shape == "fb":
                rec("fused_gcs_ccorr", timeit(lambda: K.fused_gcs("ccorr", ent, cp.xi, rel, cp.yi, cp.scal, cp.by_seg, 2 * N), 3, 1), 0, 2 * E * D * D)
    if want("linear"):
        W = rnd(D, D) / D ** 0.5; b = rnd(D); xe = x[:E]
        fl = 2 * E * D * D
        with torch.no_grad():
            rec("linear_fwd", timeit(lambda: K._Linear.forward(_Ctx(), xe, W, b, 1), args.reps), 0, fl)
            rec("torch_addmm(rocBLAS)", timeit(lambda: torch.addmm(b, xe, W.t()), args.reps), 0, fl)
        xs = xe.clone().requires_grad_(True); Ws = W.clone().requires_grad_(True)
        out = K.linear(xs, Ws, b, None)
        gy = rnd(E, D)
        rec("linear_bwd_input", timeit(lambda: torch.autograd.grad(out, xs, gy, retain_graph=True), args.reps), 0, fl)
        rec("linear_bwd_weight", timeit(lambda: torch.autograd.grad(out, Ws, gy, retain_graph=True), args.reps), 0, fl)
    if want("wgrad"):
        from mr_gnas_amd._lib import call, ptr, stream_of
        lib = _lib.load()
        for rows, K1, K2 in ((M, D, D), (M, D, 0), (E // 2, D, D), (N, D, D)):
            gy, x1, x2 = rnd(rows, D), rnd(rows, K1), (rnd(rows, K2) if K2 else None)
            gW, gb = torch.empty(D, K1 + K2, device=dev), torch.empty(D, device=dev)
            ws = torch.empty(int(lib.mrg_linear_bwd_weight_workspace_bytes(rows, K1 + K2, D)), dtype=torch.uint8, device=dev)
            fn = lambda: call("mrg_linear_bwd_weight", (ptr(gy), ptr(x1), ptr(x2), ptr(gW), ptr(gb), ptr(ws), rows, K1, K2, D, stream_of(gy)))
            rec(f"wgrad rows={rows} K={K1}+{K2}", timeit(fn, args.reps), 0, 2 * rows * (K1 + K2) * D)
            gx = torch.empty(rows, K1, device=dev); W = rnd(D, K1 + K2)
            wt = torch.empty(int(lib.mrg_linear_bwd_input_workspace_bytes(K1, D)), dtype=torch.uint8, device=dev)
            fn2 = lambda: call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(wt), rows, K1, D, K1 + K2, 0, stream_of(gy)))
            rec(f"bwd_input rows={rows} K={K1}", timeit(fn2, args.reps), 0, 2 * rows * K1 * D)
    print(json.dumps({"shape": args.shape, "N": N, "E": E, "D": D, "kernels": res}))


class _Ctx:
    """Minimal stand-in for an autograd ctx when timing a Function.forward directly."""
    needs_input_grad = (False,) * 16

    def save_for_backward(self, *a):
        pass


if __name__ == "__main__":
    main()
