#!/bin/bash
# quick serialized kernel trace of the headline bench: per-kernel median durations.  Usage: tools/gpu_trace_quick.sh <tag> [bench args]
set -u
tag=${1:-q}; shift || true
export TMPDIR=/tmp
rm -rf gpurun_out/$tag; mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$tag -- python3 bench.py --no-cpu-baseline --no-e2e --steps 3 --warmup 1 "$@" > gpurun_out/$tag/bench.log 2>&1
python3 - gpurun_out/$tag <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    if n.startswith('at::') or n.startswith('__amd'): continue
    d[n[:44]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    v2=sorted(v); print(f"{n:44s} n={len(v):4d} median {v2[len(v2)//2]:8.1f} p25 {v2[len(v2)//4]:8.1f} p75 {v2[3*len(v2)//4]:8.1f} us")
PY
