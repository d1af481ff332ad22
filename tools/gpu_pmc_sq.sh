#!/bin/bash
# SQ activity counters of selected kernels (KERNEL_RE) for one bench workload: where the wave cycles go.
# Usage: KERNEL_RE='lkIteration' tools/gpu_pmc_sq.sh <tag> [bench args]
set -u
tag=${1:-sq}; shift || true
export TMPDIR=/tmp
out=gpurun_out/$tag; rm -rf "$out"; mkdir -p "$out"
B="python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0 $*"
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_ANY SQ_LEVEL_WAVES SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "${KERNEL_RE:-accumulate}" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$out/p$i.log"; exit 1; }
done
python3 - "$out" <<'PY'
import csv,glob,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+'/p*/*/*_counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,c in acc.items():
    print(k)
    for n in sorted(c): print(f"    {n:24s} mean {sum(c[n])/len(c[n]):.4g}  (n={len(c[n])})")
PY
