#!/bin/bash
# A/B of an environment knob by per-kernel medians of a SERIALIZED kernel trace (cfg.asyncFuse = 0) and by the default bench:
#   AB_VAR=MFSR_LK_XCD AB_VALUES="0 1" tools/gpu_ab_env_trace.sh <tag> [kernel regex]
set -u
tag=${1:-abenv}; re=${2:-.}
export TMPDIR=/tmp
for round in 1 2; do
  for v in ${AB_VALUES}; do
    d=gpurun_out/$tag/r${round}_$v
    rm -rf $d; mkdir -p $d
    env ${AB_VAR}=$v rocprofv3 --kernel-trace --output-format csv -d $d -- python3 bench.py --no-cpu-baseline --no-e2e --no-isolated --no-async-fuse --steps 3 --warmup 1 ${BENCH_ARGS:-} > $d/bench.log 2>&1
    python3 - $d "${AB_VAR}=$v" "$re" <<'PY'
import csv,glob,sys,collections,re,json
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
d=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','')
    if n.startswith('at::') or n.startswith('__amd'): continue
    d[n[:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print(f"== {sys.argv[2]} (serialized trace)")
for n,v in sorted(d.items(), key=lambda kv:-sum(kv[1])):
    if not re.search(sys.argv[3], n): continue
    v2=sorted(v); print(f"   {n:40s} n={len(v):4d} median {v2[len(v2)//2]:8.1f} sum/burst {sum(v)/4:9.1f} us")
PY
    env ${AB_VAR}=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-isolated ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   default bench: ${AB_VAR}=$v', d['ms_per_step'], 'ms per burst')"
  done
done
