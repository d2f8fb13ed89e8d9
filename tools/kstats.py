#!/usr/bin/env python3
"""Print rows of a rocprofv3 kernel_stats.csv whose kernel name contains any of the given substrings: name, calls, avg us, min us, max us."""
import csv, sys
path, pats = sys.argv[1], sys.argv[2:]
for r in csv.DictReader(open(path)):
    if any(p in r["Name"] for p in pats):
        print(f'{r["Name"][:60]:60s} calls {r["Calls"]:>6s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  max {float(r["MaxNs"])/1e3:9.1f}')
