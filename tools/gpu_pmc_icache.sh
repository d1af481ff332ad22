#!/bin/bash
# instruction-cache counters of the warp+fuse tile kernel in the headline bench.  Usage: tools/gpu_pmc_icache.sh <tag>
set -u
tag=${1:-pmc_icache}
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --list-avail > "$out/avail.txt" 2>&1 || rocprofv3 -L > "$out/avail.txt" 2>&1
grep -o "SQC_[A-Z_0-9]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQ_[A-Z_]*IFETCH[A-Z_]*" "$out/avail.txt" | sort -u > "$out/names.txt"
cat "$out/names.txt" | tr '\n' ' '; echo
B="python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0"
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "accumulate2xTile" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1 \
    || { echo "pass $i failed"; tail -5 "$out/p$i.log"; }
done
python3 - "$out" <<'PY'
import collections, csv, glob, sys
per = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/*/*counter_collection.csv"):
    d = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        d[(r["Counter_Name"], int(r["Dispatch_Id"]))] += float(r["Counter_Value"])
    for (c, i), v in d.items():
        per[c].append(v)
for c, v in sorted(per.items()):
    print(f"{c:36s} n={len(v):3d} mean {sum(v)/len(v):.4g}")
PY
