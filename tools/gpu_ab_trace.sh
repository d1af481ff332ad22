#!/bin/bash
# A/B of library builds by per-kernel medians of a SERIALIZED kernel trace (cfg.asyncFuse = 0: one stream, no overlap):
#   AB_LIBS="multi_frame_super_resolution_amd/lib/libmfsr_hip.so build_ab/libmfsr_x.so" tools/gpu_ab_trace.sh <tag> [kernel regex]
set -u
tag=${1:-ab}; re=${2:-.}
export TMPDIR=/tmp
for round in 1 2; do
  i=0
  for l in ${AB_LIBS}; do
    i=$((i+1))
    d=gpurun_out/$tag/r${round}_$i
    rm -rf $d; mkdir -p $d
    MFSR_LIB=$PWD/$l rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --no-e2e --no-isolated --no-async-fuse --steps 3 --warmup 1 ${BENCH_ARGS:-} > $d/bench.log 2>&1
    python3 - $d "$l" "$re" <<'PY'
import csv,glob,sys,collections,re,json
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    if n.startswith('at::') or n.startswith('__amd'): continue
    d[n[:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
ms=None
for ln in open(sys.argv[1]+'/bench.log'):
    if ln.startswith('{'): ms=json.loads(ln)['ms_per_step']
print(f"== {sys.argv[2]}: {ms} ms per burst (serialized, under the profiler)")
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if not re.search(sys.argv[3], n): continue
    v2=sorted(v); print(f"   {n:40s} n={len(v):4d} median {v2[len(v2)//2]:8.1f} sum/burst {sum(v)/4:9.1f} us")
PY
  done
done
