#!/bin/bash
# PMC passes of the warp+fuse launch with forced certainty masks (tools/fuse_sat_ab.py): per dispatch of k_accumulate2xTile, in
# launch order (saturated / general / saturated / general, 3 + SAT_REPS launches each), instruction counts and SQ activity.
# Usage: tools/gpu_pmc_fuse_sat.sh <tag>
set -u
tag=${1:-pmc_fuse_sat}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp SAT_STATS=0 SAT_REPS=4
i=0
for grp in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM SQ_INSTS_SALU SQ_IFETCH SQ_ACTIVE_INST_SCA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "accumulate2xTile" --output-format csv -d "$out/p$i" -- python3 tools/fuse_sat_ab.py > "$out/p$i.log" 2>&1 \
    || { echo "pass $i failed"; tail -3 "$out/p$i.log"; exit 1; }
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
per = collections.defaultdict(dict)   # dispatch id -> counter -> value
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    rows = list(csv.DictReader(open(f)))
    ids = sorted({int(r["Dispatch_Id"]) for r in rows})
    rank = {d: i for i, d in enumerate(ids)}
    for r in rows:
        d = per[rank[int(r["Dispatch_Id"])]]
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
n = len(per)
# launch order: [pipeline launches...] then 4 variants x (3 + 4); take the last 28
ids = sorted(per)[-28:]
groups = {"ones": ids[0:7] + ids[14:21], "general": ids[7:14] + ids[21:28]}
with open(out + "/summary.txt", "w") as fo:
    for g, lst in groups.items():
        names = sorted({c for i in lst for c in per[i]})
        line = f"{g}: " + "  ".join(f"{c} {sum(per[i].get(c, 0) for i in lst) / len(lst):.4g}" for c in names)
        print(line)
        fo.write(line + "\n")
PY
