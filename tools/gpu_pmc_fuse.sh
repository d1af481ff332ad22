#!/bin/bash
# SQ-only PMC passes focused on the warp+fuse kernel (one counter group per pass; TA/TCP/TCC
# groups crash the profiler on this pool and are not collected).  Usage: tools/gpu_pmc_fuse.sh <tag>
set -u
tag=${1:-fuse}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --steps 1 --warmup 0"
i=0
for grp in \
 "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY" \
 "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM" \
 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_BUSY_CYCLES SQ_CYCLES SQ_INSTS_SALU" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-include-regex "${KERNEL_RE:-accumulate2x}" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1 || { echo "pass $i failed"; tail -5 "$out/p$i.log"; exit 1; }
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(out+'/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for k,v in sorted(agg.items()):
        line=f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.6g}"
        print(line); fo.write(line+'\n')
PY
