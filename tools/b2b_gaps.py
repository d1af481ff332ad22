#!/usr/bin/env python3
"""Steady-state burst of tools/b2b_trace.py's kernel trace: per compute-queue kernel of the second-to-last burst its start, duration
and the idle gap before it; totals of busy time and gaps."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")) for r in csv.DictReader(open(f)))
rows = [r for r in rows if r[2].startswith("k_")]
fin = [i for i, r in enumerate(rows) if "k_deBayerFused" in r[2]]          # one per burst (reference products)
a, b = fin[-3], fin[-2]
seg = rows[a:b]
t0 = seg[0][0]
busy = sum(e - s for s, e, _ in seg)
span = seg[-1][1] - t0 if False else rows[b][0] - t0
print(f"burst period {span / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us (sum of durations; overlaps counted twice), {len(seg)} launches")
prev_end = seg[0][0]
gaps = []
for s, e, n in seg:
    g = s - prev_end
    if g > 15000:
        gaps.append((g, (s - t0), n))
    prev_end = max(prev_end, e)
print("gaps > 15 us before a kernel:", sum(g for g, _, _ in gaps) / 1e3, "us in", len(gaps))
for g, at, n in gaps:
    print(f"   {g / 1e3:7.1f} us gap at {at / 1e3:8.1f} us before {n[:60]}")
import collections
c = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    c[n[:36]][0] += 1; c[n[:36]][1] += e - s
for n, (k, d) in sorted(c.items(), key=lambda kv: -kv[1][1]):
    print(f"   {n:36s} x{k:3d} {d / 1e3:8.1f} us")
