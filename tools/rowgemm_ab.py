#!/usr/bin/env python3
"""A/B of the split-core row GEMM kernels at the headline shapes, interleaved rounds in ONE process (cdna_hip_programming.md
rule 24): rowgemm_x3q_k (16 x 16 x 32 tiles, three workgroups per CU; mrg_gemm_set_q(1)) against rowgemm_x3s_k (set_q(0)).

    python tools/rowgemm_ab.py [--rows 558771] [--rounds 7] [--reps 20]

Entry points timed (each is one launch of the row GEMM + the tiny weight split):
  linear      mrg_linear_fwd            rows x 200 x 200, bias + ReLU                       (a_max / a_mean's Linear)
  bwd_input   mrg_linear_bwd_input      rows x 200 x 200, plain store                       (input gradients)
  bwd_acc     mrg_linear_bwd_input      the same, accumulating into gX                      (EPI_ACCUM)
  gate3       mrg_dense_filter_fwd3     kind 0, K = 400, gate only (out == NULL)            (f_dense_comp forward, three segments)
  scale3      mrg_dense_filter_fwd3     kind 1, K = 400, stored output                      (f_comp forward)
  pair3       mrg_linear_bwd_input3_pair  K = 400 -> 200 columns                            (the two candidates' input gradient)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import _lib  # noqa: E402
from mr_gnas_amd._lib import call, ptr, ptr_array, stream_of  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=558771)
    ap.add_argument("--dim", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lib = _lib.load()
    dev = "cuda"
    M, D = a.rows, a.dim
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
    s, s_in, gy, gy2 = rnd(M, D), rnd(M, D), rnd(M, D), rnd(M, D)
    W = rnd(D, D) / D ** 0.5
    W2 = [rnd(D, 2 * D) / (2 * D) ** 0.5 for _ in range(3)]
    W2b = [rnd(D, 2 * D) / (2 * D) ** 0.5 for _ in range(3)]
    b = rnd(D)
    b3 = [rnd(D) for _ in range(3)]
    norm = torch.rand(M, device=dev, generator=gen) + 0.1
    out, gate, gx = torch.empty(M, D, device=dev), torch.empty(M, D, device=dev), torch.zeros(M, D, device=dev)
    b0, b1 = (M - 14541) // 2, M - 14541
    ws = torch.empty(max(16, int(lib.mrg_gemm_workspace_bytes(2 * D, D)) * 4), dtype=torch.uint8, device=dev)
    ws3 = torch.empty(max(16, int(lib.mrg_dense_filter3_workspace_bytes(D, 2 * D))), dtype=torch.uint8, device=dev)
    wsp = torch.empty(max(16, int(lib.mrg_linear_bwd_input3_pair_workspace_bytes(D, D))), dtype=torch.uint8, device=dev)
    st = stream_of(out)
    cases = {
        "linear": (lambda: call("mrg_linear_fwd", (ptr(s), ptr(W), ptr(b), ptr(out), ptr(ws), M, D, D, 1, st)), 2.0 * M * D * D),
        "bwd_input": (lambda: call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), M, D, D, D, 0, st)), 2.0 * M * D * D),
        "bwd_acc": (lambda: call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), M, D, D, D, 1, st)), 2.0 * M * D * D),
        "gate3": (lambda: call("mrg_dense_filter_fwd3", (0, ptr(s), ptr(s_in), ptr_array(W2), ptr_array(b3), ptr(norm), 1 / 3, 1 / 3, None, ptr(gate),
                                                         ptr(ws3), b0, b1, M, D, st)), 4.0 * M * D * D),
        "scale3": (lambda: call("mrg_dense_filter_fwd3", (1, ptr(s), ptr(s_in), ptr_array(W2b), ptr_array([None, None, None]), ptr(norm), 1 / 3, 1.0,
                                                          ptr(out), None, ptr(ws3), b0, b1, M, D, st)), 4.0 * M * D * D),
        "pair3": (lambda: call("mrg_linear_bwd_input3_pair", (ptr(gy), ptr(gy2), ptr_array([w[:, :D] for w in W2]), ptr_array([w[:, :D] for w in W2b]),
                                                              ptr(gx), ptr(wsp), b0, b1, M, D, D, 2 * D, 0, st)), 4.0 * M * D * D),
    }
    only = set(filter(None, a.only.split(",")))
    res = {}
    for name, (fn, flops) in cases.items():
        if only and name not in only:
            continue
        t = {0: [], 2: []}
        for q in (2, 0):
            lib.mrg_gemm_set_q(q)
            for _ in range(3):
                fn()
        torch.cuda.synchronize()
        for r in range(a.rounds):
            for q in ((2, 0) if r % 2 == 0 else (0, 2)):
                lib.mrg_gemm_set_q(q)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                t[q].append(e0.elapsed_time(e1) / a.reps)
        lib.mrg_gemm_set_q(1)
        rec = {}
        for q, nm in ((2, "x3q"), (0, "x3s")):
            med, mn = float(np.median(t[q])), float(np.min(t[q]))
            rec[nm] = {"ms_median": round(med, 4), "ms_min": round(mn, 4), "tflops_f32eq": round(flops / med / 1e9, 1), "frac_of_416.7": round(flops / med / 1e9 / 416.7, 3)}
        rec["x3q_over_x3s"] = round(rec["x3q"]["ms_median"] / rec["x3s"]["ms_median"], 3)
        res[name] = rec
        print(name, json.dumps(rec), flush=True)
    print(json.dumps({"rows": M, "dim": D, "results": res}))


if __name__ == "__main__":
    main()
