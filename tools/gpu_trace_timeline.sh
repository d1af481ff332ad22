#!/bin/bash
# overlapped kernel timeline of the last burst of the headline bench: start offset, duration and queue of every launch.
# Usage: tools/gpu_trace_timeline.sh <tag> [bench args]
set -u
tag=${1:-tl}; shift || true
export TMPDIR=/tmp
rm -rf gpurun_out/$tag; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > gpurun_out/$tag/bench.log 2>&1
python3 - gpurun_out/$tag <<'PY' > gpurun_out/$tag/timeline.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
rows=[]
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    if n.startswith('at::') or n.startswith('__amd'): continue
    rows.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),n[:40],r.get('Queue_Id','?')))
rows.sort()
# bursts start at k_prepareFrame launches with a gap before: take the last set-reference marker = last 'k_tileSq' or Derivatives2
starts=[i for i,r in enumerate(rows) if 'Derivatives2' in r[2]]
i0=starts[-1]
# walk back to the first launch of that burst (prepare of the reference): up to 6 launches earlier
i0=max(0,i0-8)
t0=rows[i0][0]
for s,e,n,q in rows[i0:]:
    print(f"{(s-t0)/1e3:9.1f} +{(e-s)/1e3:8.1f} us  q{q:>3s}  {n}")
PY
tail -150 gpurun_out/$tag/timeline.txt
