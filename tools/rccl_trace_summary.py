#!/usr/bin/env python3
"""Where did the RCCL kernels of tools/rccl_under_fuse.py sit against the warp+fuse launches?  Reads the kernel trace of
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 tools/rccl_under_fuse.py` and prints, for every RCCL
kernel that overlaps a k_accumulate2xTile dispatch: its start relative to the fuse dispatch's start, its duration, and the
fuse dispatch's duration; plus the duration of the RCCL kernels that ran with nothing beside them."""
import csv
import glob
import statistics
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
fuse = [(s, e) for n, s, e, q in rows if "k_accumulate2xTile" in n]
rccl = [(n, s, e, q) for n, s, e, q in rows if "nccl" in n.lower() or "rccl" in n.lower()]
print(f"{len(fuse)} warp+fuse dispatches, {len(rccl)} RCCL kernels ({sorted(set(n[:60] for n, *_ in rccl))})")
inside, alone = [], []
for n, s, e, q in rccl:
    hit = [(fs, fe) for fs, fe in fuse if s < fe and e > fs]
    if hit:
        fs, fe = hit[0]
        inside.append(((s - fs) / 1e3, (e - s) / 1e3, (fe - fs) / 1e3, (e - fe) / 1e3, q))
    else:
        alone.append((e - s) / 1e3)
if alone:
    print(f"alone: {len(alone)} kernels, median {statistics.median(alone):.1f} us")
by_q = {}
for st, du, fd, tail, q in inside:
    by_q.setdefault(q, []).append((st, du, fd, tail))
for q, v in by_q.items():
    print(f"queue {q}: {len(v)} RCCL kernels overlapping a fuse dispatch: started {statistics.median(x[0] for x in v):.0f} us after its start (median), "
          f"ran {statistics.median(x[1] for x in v):.1f} us (median; max {max(x[1] for x in v):.1f}), fuse dispatch {statistics.median(x[2] for x in v):.0f} us; "
          f"ended {statistics.median(-x[3] for x in v):.0f} us BEFORE the fuse dispatch did (median; min {min(-x[3] for x in v):.0f})")
