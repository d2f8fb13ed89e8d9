#!/usr/bin/env python3
"""Lab: where one step's time goes, from a tools/trace_kernels.sh trace ('.' pattern).
Splits the dispatch list into steps at the gaps > 1 ms... (bench.py synchronises between steps), then for the last full
step prints: wall span, union of busy intervals (GPU not idle), sum of kernel durations, and the top kernels by time."""
import collections, re, sys

rows = []
for line in open(sys.argv[1]):
    m = re.match(r"\s*(\d+) (\d+)\s+([\d.]+) us grid\s+(\S+) (.*?)\s+after ", line)
    if m:
        rows.append((int(m.group(2)), float(m.group(3)) * 1e3, m.group(5).strip()))
rows.sort()
# steps: split where the gap between consecutive starts exceeds 2 ms
steps, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - (a[0] + a[1]) > 2e6:
        steps.append(cur); cur = []
    cur.append(b)
steps.append(cur)
print("segments:", [(len(s), round((s[-1][0] + s[-1][1] - s[0][0]) / 1e6, 2)) for s in steps])
which = int(sys.argv[2]) if len(sys.argv) > 2 else max(range(len(steps)), key=lambda i: len(steps[i]))
st = steps[which]
t0, t1 = st[0][0], max(s + d for s, d, _ in st)
busy, end = 0.0, t0
for s, d, _ in st:
    if s + d > end:
        busy += s + d - max(s, end); end = s + d
tot = sum(d for _, d, _ in st)
print(f"segment {which}: {len(st)} kernels, wall {(t1 - t0) / 1e6:.2f} ms, GPU busy (union) {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms, "
      f"sum of kernel durations {tot / 1e6:.2f} ms")
agg = collections.defaultdict(lambda: [0, 0.0])
for _, d, n in st:
    k = re.sub(r"\(.*", "", n)[:64]
    agg[k][0] += 1; agg[k][1] += d
for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print(f"  {k:64s} {c:5d} {d / 1e6:8.2f} ms {100 * d / tot:5.1f}%  avg {d / c / 1e3:8.1f} us")
