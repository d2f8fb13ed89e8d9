#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md prescribes: the
two counters do not fit one pass) into HBM-side bytes PER DISPATCH of every device kernel and of the C-ABI
entry points, with the guide's gfx950 corrections: both counters are in KiB; FETCH_SIZE counts a 128-B request as
64 B for wide (16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is exact.

    python tools/traffic_from_pmc.py gpurun_out/TAG/pmc_fetch gpurun_out/TAG/pmc_write profiles/r2_traffic.json [D M]

Counter rows are first summed per Dispatch_Id (a dispatch may be reported in several rows), then averaged over the
dispatches of a kernel; kernels are keyed by (name, grid size) so the two north-star passes (FB15k-237 and C5
shapes launch the same template) stay apart.  Self-check: `compose_fwd_k` must come out at 12*D*M bytes.
"""
import collections
import csv
import glob
import json
import re
import sys

# C-ABI entry point -> substrings of the device kernels ONE call launches (each once); missing kernels are skipped
ENTRY_KERNELS = {
    "mrg_linear_bwd_weight": ["wgrad_x3_k", "wgrad_reduce3_k"],
    "mrg_linear_bwd_input": ["rowgemm_x3_k<7, 2, 0"],
    "mrg_dense_filter_fwd": ["rowgemm_x3_k<7, 2, 1"],
    "mrg_sum_buffers": ["sum_k"],
    "mrg_distmult_score": ["distmult_k"],
    "mrg_mix_bwd_apply": ["mix_bwd_apply_k"],
    "mrg_mix_fwd": ["mix_fwd_k"],
    "mrg_mix_colstats": ["mix_colstats_k"],
    "mrg_mix_stats_coef": ["mix_colstats_k", "mix_reduce_finalize_fwd_k"],
    "mrg_mix_bwd_reduce": ["mix_bwd_reduce_k"],
    "mrg_gate_fwd": ["gate_fwd_k"],
    "mrg_gate_bwd": ["gate_bwd_k"],
    "mrg_compose_fwd": ["compose_fwd_k"],
    "mrg_compose_bwd": ["compose_bwd_k"],
    "mrg_seg_reduce_bwd": ["seg_bwd_k"],
    "mrg_dense_filter_dz": ["dense_dz_k"],
    # the three-segment entry points run the same kernels over all M rows: their dispatches are the ones with the largest grid
    "mrg_dense_filter_fwd3": ["rowgemm_x3_k<7, 2, 1@max|rowgemm_x3_k<7, 2, 2@max"],      # a|b: one or the other per call
    "mrg_linear_bwd_input3": ["rowgemm_x3_k<7, 2, 0@max|rowgemm_x3_k<7, 2, 3@max"],
    "mrg_linear_bwd_weight3": ["wgrad_x3_k@max", "wgrad_reduce3_k"],
    "mrg_dense_filter_dz3": ["dense_dz_k@max"],
    "mrg_linear_relu_segmax_fwd": ["rowgemm_x3_k<7, 2, 4", "segmax_finalize_k"],
    "mrg_linear_relu_segsum_fwd": ["rowgemm_x3_k<7, 2, 5"],
}
# round 3: the default split-core kernel is rowgemm_x3s_k<NT, EPI, DUAL> (EPI 0 bias/act, 1 gate, 2 scale, 3 accumulate, 4 segmax, 5 segsum)
ENTRY_KERNELS.update({
    "mrg_linear_bwd_input": ["rowgemm_x3s_k<7, 0, false"],
    "mrg_dense_filter_fwd": ["rowgemm_x3s_k<7, 1, false"],
    "mrg_dense_filter_fwd3": ["rowgemm_x3s_k<7, 1,@max|rowgemm_x3s_k<7, 2,@max"],
    "mrg_linear_bwd_input3": ["rowgemm_x3s_k<7, 0, false@max|rowgemm_x3s_k<7, 3, false@max"],
    "mrg_linear_bwd_input3_pair": ["rowgemm_x3s_k<7, 3, true@max|rowgemm_x3s_k<7, 0, true@max"],
    "mrg_linear_relu_segmax_fwd": ["rowgemm_x3s_k<7, 4", "segmax_finalize_k"],
    "mrg_linear_relu_segsum_fwd": ["rowgemm_x3s_k<7, 5"],
    # round 5: the 16 x 16 x 32 kernel takes the K = 400 products (rowgemm_x3q_k<EPI, DUAL>), a_max's input gradient has no GEMM
    "mrg_dense_filter_fwd3": ["rowgemm_x3q_k<1, true>@max|rowgemm_x3q_k<2, true>@max|rowgemm_x3s_k<7, 1,@max|rowgemm_x3s_k<7, 2,@max"],
    "mrg_linear_bwd_input3_pair": ["rowgemm_x3q_k<3, true>@max|rowgemm_x3q_k<0, true>@max|rowgemm_x3s_k<7, 3, true@max"],
    "mrg_segmax_bwd_input": ["segmax_bwd_gx_k"],
    "mrg_zero_stats_coef": ["zero_colstats_k", "mix_reduce_finalize_fwd_k"],
    "mrg_zero_fwd": ["zero_fwd_k"],
    "mrg_zero_bwd_reduce": ["zero_bwd_reduce_k"],
    "mrg_zero_bwd_apply": ["zero_bwd_apply_k"],
    "mrg_span_gcs": ["span_gcs_k<4, 64, 1, 2|span_gcs_k<4, 64, 1, 1|span_gcs_k<4, 64, 1, 3"],
    "mrg_seg_reduce_bwd_bits": ["seg_bwd_bits_k"],
    # round 3, later: the weight gradient that splits every fragment once per workgroup; f_sparse_comp as a row factor
    "mrg_linear_bwd_weight": ["wgrad_x3v_k", "wgrad_reduce3_k"],
    "mrg_linear_bwd_weight3": ["wgrad_x3v_k@max", "wgrad_reduce3_k"],
    "mrg_gate_row_fwd": ["gate_row_fwd_k"],
    "mrg_gate_row_bwd": ["gate_row_bwd_k"],
    "mrg_sum_rows_gather": ["sum_rows_gather_k"],
})
NORTH_STAR = "span_gcs_k<4, 64, 1, 0,"            # MODE = SUB only runs in bench.py's north-star passes (last parameter: prefetch-depth override)


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"^mrg::", "", name)
    return name.split("(")[0][:100]


def per_dispatch(dirpath, counter):
    """{(kernel, grid): [bytes of each dispatch]} in KiB as reported."""
    disp = {}
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r["Counter_Name"] != counter:
                    continue
                key = (f, r["Dispatch_Id"])
                name, grid, val = disp.get(key, (r["Kernel_Name"], int(r["Grid_Size"]), 0.0))
                disp[key] = (name, grid, val + float(r["Counter_Value"]))
    out = collections.defaultdict(list)
    for name, grid, val in disp.values():
        out[(short(name), grid)].append(val)
    return out


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch = per_dispatch(fetch_dir, "FETCH_SIZE")
    write = per_dispatch(write_dir, "WRITE_SIZE")
    mean = lambda xs: sum(xs) / len(xs) if xs else 0.0
    kern = {}
    for key in sorted(set(fetch) | set(write)):
        fb, wb = 2.0 * 1024 * mean(fetch.get(key, [])), 1024.0 * mean(write.get(key, []))
        kern[key] = {"dispatches": len(fetch.get(key, write.get(key, []))), "fetch_bytes": int(fb), "write_bytes": int(wb),
                     "bytes_per_dispatch": int(fb + wb)}
    res = {"source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, with --kernel-trace) over "
                     "`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`",
           "correction": "bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 per dispatch (MI355X_MICROARCH.md: counters in KiB; "
                         "gfx950 FETCH_SIZE tallies 128-B requests as 64 B)",
           "note": "FETCH_SIZE counts requests that leave L2, Infinity-Cache hits included: it is an upper bound of DRAM reads",
           "per_launch_bytes": {}, "device_kernels": {}}
    for (name, grid), v in kern.items():
        if name.startswith(("mrg::", "span_", "seg_", "mix_", "gate_", "compose_", "rowgemm", "wgrad", "sum_k", "dense_", "distmult",
                            "gather_", "bsplit", "ordered_reduce", "plan_", "graph_", "sample_", "score_", "rank_", "linrelu", "zero_")) or "mrg" in name:
            res["device_kernels"][f"{name} [grid {grid}]"] = v
    for entry, subs in ENTRY_KERNELS.items():
        total, found = 0.0, False
        for alts in subs:
            keys = []
            for sub in alts.split("|"):
                sub, _, pick = sub.partition("@")
                ks = [k for k in kern if sub in k[0]]
                if ks and pick == "max":
                    top = max(k[1] for k in ks)
                    ks = [k for k in ks if k[1] == top]
                keys += ks
            if not keys:
                continue
            found = True
            w = sum(kern[k]["dispatches"] for k in keys)
            total += sum(kern[k]["bytes_per_dispatch"] * kern[k]["dispatches"] for k in keys) / max(w, 1)
        if found:
            res["per_launch_bytes"][entry] = int(total)
    ns = sorted((k for k in kern if NORTH_STAR in k[0]), key=lambda k: k[1])
    if ns:
        res["per_launch_bytes"]["north_star:fb15k237"] = kern[ns[0]]["bytes_per_dispatch"]
        if len(ns) > 1:
            res["per_launch_bytes"]["north_star:c5_synthetic10m"] = kern[ns[-1]]["bytes_per_dispatch"]
    # HBM-side bytes of ONE step: every dispatch of the profiled command except the north-star / calibration passes, divided by the
    # number of steps it ran (bench.py --steps 1 --warmup 1: warm-up + instrumented + timed = 3)
    steps = int(sys.argv[6]) if len(sys.argv) >= 7 else 3
    skip = (NORTH_STAR, "compose_fwd_k")
    tot = sum(v["bytes_per_dispatch"] * v["dispatches"] for (name, grid), v in kern.items() if not any(x in name for x in skip))
    res["step_total"] = {"steps_in_profile": steps, "bytes_per_step": int(tot / steps), "GB_per_step": round(tot / steps / 1e9, 1),
                         "excluded": "north-star passes (span_gcs_k MODE sub) and the compose_fwd_k calibration launches"}
    print("HBM-side bytes per step: %.1f GB (%d steps in the profile)" % (tot / steps / 1e9, steps))
    if len(sys.argv) >= 6:                                         # self-check against a known byte count
        D, M = int(sys.argv[4]), int(sys.argv[5])
        got, want = res["per_launch_bytes"].get("mrg_compose_fwd"), 12 * D * M
        res["calibration"] = {"kernel": "compose_fwd_k", "expected_bytes": want, "measured_bytes": got,
                              "ratio": round(got / want, 4) if got else None}
        print("calibration compose_fwd_k:", res["calibration"])
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in res["per_launch_bytes"].items():
        print(f"{k:34s} {v / 1e6:10.1f} MB per launch")


if __name__ == "__main__":
    main()
