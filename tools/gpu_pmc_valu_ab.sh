#!/bin/bash
# VALU wave-instructions and busy cycles per k_accumulate2xTile dispatch of the headline bench, for the default library and for
# the one named by $1 (an A/B build).  Usage: tools/gpu_pmc_valu_ab.sh <other lib> <tag>
set -u
other=$1; tag=${2:-pmc_valu_ab}
out=gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex "accumulate2xTile" --output-format csv -d "$out/a" -- $B > "$out/a.log" 2>&1 || { tail -3 "$out/a.log"; exit 1; }
export MFSR_LIB=$other
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-include-regex "accumulate2xTile" --output-format csv -d "$out/b" -- $B > "$out/b.log" 2>&1 || { tail -3 "$out/b.log"; exit 1; }
python3 - "$out" <<'PY'
import collections, csv, glob, sys
for v in "ab":
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(f"{sys.argv[1]}/{v}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            per[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per)
    print(v, "dispatches", len(ids), "VALU per dispatch:", " ".join(f"{per[i]['SQ_INSTS_VALU']/1e8:.3f}" for i in ids[-8:]),
          "| busy:", " ".join(f"{per[i]['SQ_BUSY_CYCLES']/1e7:.3f}" for i in ids[-8:]))
PY
