#!/usr/bin/env python3
"""Per-kernel sums of the SQ counters of tools/pmc_sq.sh (rocprofv3 counter_collection.csv) for the kernels that dominate the step:
share of wave cycles spent parked (s_waitcnt / barrier), stalled at issue (matrix pipe / dependencies) and issuing; the matrix pipe's
busy cycles (SQ_VALU_MFMA_BUSY_CYCLES = 32 per v_mfma_f32_32x32x16_bf16, summed over the 1024 SIMDs) against the launches' duration
(dispatch timestamps; the pipe's utilisation is given for a 2.4 GHz clock, i.e. it is a LOWER bound when the chip clocks lower under
load); LDS bank-conflict cycles per wave cycle."""
import csv, glob, sys, collections
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
calls, dur = collections.Counter(), collections.Counter()
seen = set()
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0].replace("void mrg::", "").replace("mrg::", "")
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], name)
        if key not in seen:
            seen.add(key)
            calls[name] += 1
            dur[name] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
rows = sorted(acc.items(), key=lambda kv: -dur[kv[0]])[:16]
print(f"{'kernel':40s} {'calls':>5s} {'ms total':>9s} {'parked':>7s} {'issue stall':>11s} {'issuing':>8s} {'MFMA pipe busy @2.4GHz':>22s} {'LDS conflict/wave cyc':>21s}")
for name, c in rows:
    wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    cyc = dur[name] * 2.4 * 1024            # ns * GHz * SIMDs
    print(f"{name[:40]:40s} {calls[name]:5d} {dur[name] / 1e6:9.3f} {c.get('SQ_WAIT_ANY', 0) / wc:7.3f} {c.get('SQ_WAIT_INST_ANY', 0) / wc:11.3f} "
          f"{c.get('SQ_ACTIVE_INST_ANY', 0) / wc:8.3f} {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / cyc:22.3f} {c.get('SQ_LDS_BANK_CONFLICT', 0) / wc:21.4f}")
