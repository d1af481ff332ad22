#!/usr/bin/env python3
"""List the kernels of the LAST burst of a kernel trace in start order: name, stream/queue, start (us from the burst's first
kernel), duration.  python3 tools/trace_kernels.py <rocprofv3 output dir> [first-kernel regex]"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
first = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"k_prepareFrameFused|k_deBayersSubSample3")
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""),
         r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
# a burst starts with the reference frame's prepare kernel after a finish kernel / a long gap
starts = [i for i, r in enumerate(rows) if first.search(r[2]) and (i == 0 or rows[i][0] - rows[i - 1][1] > 200000 or "inish" in rows[i - 1][2])]
i0 = starts[-1] if starts else 0
t0 = rows[i0][0]
for s, e, n, q in rows[i0:]:
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  q{q}  {n[:70]}")
