#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, with --kernel-trace) into
per-launch HBM bytes of the C-ABI entry points, applying the gfx950 corrections of
MI355X_MICROARCH.md (section HBM): both counters are in KiB; FETCH_SIZE reads exactly half the bytes
of a wide (16 B/lane) coalesced stream, so it is doubled.

    python tools/traffic_from_pmc.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1_traffic.json
"""
import collections
import csv
import glob
import json
import sys

# C-ABI entry point -> substrings of the device kernels it launches (one of each per call)
ENTRY_KERNELS = {
    "mrg_linear_bwd_weight": ["wgrad_x3_k", "wgrad_reduce_k"],
    "mrg_linear_bwd_input": ["bsplit_k", "rowgemm_x3_k<7, 2, 0"],        # EPI_BIAS_ACT instances (shared with mrg_linear_fwd)
    "mrg_dense_filter_fwd": ["bsplit_k", "rowgemm_x3_k<7, 2, 1"],        # EPI_GATE
    "mrg_sum_buffers": ["sum_k"],
    "mrg_distmult_score": ["distmult_k"],
    "mrg_span_gcs": ["span_gcs_k"],
    "mrg_mix_bwd_apply": ["mix_bwd_apply_k"],
    "mrg_mix_fwd": ["mix_fwd_k"],
    "mrg_gate_fwd": ["gate_fwd_k"],
    "mrg_compose_fwd": ["compose_fwd_k"],
}


def per_kernel(dirpath, counter):
    files = glob.glob(dirpath + "/*/*counter_collection.csv")
    tot, cnt = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                cnt[r["Kernel_Name"]] += 1
    return {k: tot[k] / cnt[k] for k in tot}, cnt


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python bench.py --steps 2 --warmup 1`",
           "correction": "bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)",
           "per_launch_bytes": {}, "device_kernels": {}}
    for name in fetch:
        res["device_kernels"][name[:120]] = {"launches_profiled": nf[name], "fetch_bytes": int(2 * fetch[name] * 1024),
                                             "write_bytes": int(write.get(name, 0.0) * 1024)}
    for entry, subs in ENTRY_KERNELS.items():
        total = 0.0
        for sub in subs:
            # a C-ABI call launches each listed kernel once; several template instances may exist -> weighted mean
            names = [k for k in fetch if sub in k]
            if not names:
                continue
            w = sum(nf[k] for k in names)
            total += sum((2 * fetch[k] + write.get(k, 0.0)) * 1024 * nf[k] for k in names) / w
        if total:
            res["per_launch_bytes"][entry] = int(total)
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["per_launch_bytes"].items():
        print(f"{k:28s} {v / 1e6:10.1f} MB per launch")


if __name__ == "__main__":
    main()
