#!/bin/bash
# PMC passes focused on the warp+fuse kernel (one counter group per pass).
set -u
tag=${1:-fuse}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --steps 1 --warmup 0"
i=0
for grp in \
 "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY" \
 "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
 "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-include-regex "accumulate" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1
done
python3 - "$out" <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(list)
for f in glob.glob(out+'/p*/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in sorted(agg.items()):
    print(f"{k:40s} n={len(v):3d} mean={sum(v)/len(v):.5g}")
PY
