#!/bin/bash
# kernel trace + stats of one bench run; prints the per-kernel summary.  Usage: tools/gpu_trace.sh <tag> [bench args]
set -u
tag=${1:-trace}; shift || true
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > "$out/bench_trace.log" 2>&1
f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY' | tee "$out/kernel_stats.txt"
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>10s} {'%':>6s}")
for r in rows[:25]:
    print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")
PY
