#!/usr/bin/env python3
"""Lab: mrg_linear_bwd_input / mrg_linear_fwd at the C5 shape (rows x 256 x 256) with the eight-tile single-block row GEMM on and off
(mrg_gemm_set_wide8), dense and sparse (one non-zero per 64 rows: an a_max gradient) activations."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mr_gnas_amd import _lib  # noqa: E402
from mr_gnas_amd._lib import call, ptr, stream_of  # noqa: E402

lib = _lib.load()
dev = "cuda"
D = 256
for rows in (2_000_000, 10_000_000):
    gen = torch.Generator(device=dev).manual_seed(0)
    W = torch.randn(D, D, device=dev, generator=gen) / 16
    gx = torch.empty(rows + 1_000_000, D, device=dev)
    ws = torch.empty(int(lib.mrg_linear_bwd_input_workspace_bytes(D, D)), dtype=torch.uint8, device=dev)
    for kind in ("dense", "sparse"):
        gy = torch.randn(rows, D, device=dev, generator=gen)
        if kind == "sparse":
            gy *= (torch.rand(rows, D, device=dev, generator=gen) < 1 / 64)
        for on in (0, 1, 0, 1):
            lib.mrg_gemm_set_wide8(on)
            for acc in (0, 1):
                call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), rows, D, D, D, acc, stream_of(gx)))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    call("mrg_linear_bwd_input", (ptr(gy), ptr(W), ptr(gx), ptr(ws), rows, D, D, D, acc, stream_of(gx)))
                e1.record()
                torch.cuda.synchronize()
                ms = e0.elapsed_time(e1) / 3
                print(f"rows {rows:9d} {kind:6s} wide8={on} accumulate={acc}: {ms:7.3f} ms  {2.0 * rows * D * D / ms * 1e-9:6.1f} TF/s", flush=True)
        del gy
    del gx
lib.mrg_gemm_set_wide8(1)
