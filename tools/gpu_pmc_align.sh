#!/bin/bash
# PMC passes of the alignment kernels of the headline burst (runs on the GPU box through gpurun): per kernel and dispatch the VALU
# wave-instructions and the HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only, gfx950 corrections as in
# tools/pmc_fuse_summary.py).  Usage: tools/gpu_pmc_align.sh <tag>  -> gpurun_out/<tag>/summary.txt
set -u
tag=${1:-r03_pmc_align}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline --no-e2e --steps 1 --warmup 0"
RE="${KERNEL_RE:-lkSweep|trackTiles|robustness|prepareFrame|flowField}"
i=0
for grp in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "$RE" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1 \
    || { echo "pass $i failed"; tail -3 "$out/p$i.log"; exit 1; }
  echo "pass $i done"
done
python3 - "$out" <<'PY'
import collections, csv, glob, re, sys
out = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        per[k.group(0) if k else r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k, c in sorted(per.items()):
        n = len(c.get("SQ_INSTS_VALU", []))
        if not n:
            continue
        valu = sum(c["SQ_INSTS_VALU"]) / n
        fetch = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [1])), 1)
        write = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [1])), 1)
        hbm = 2 * fetch * 1024 + write * 1024     # gfx950: KB units, the fetch counter sees half of the reads
        busy = sum(c.get("SQ_ACTIVE_INST_VALU", [0])) / max(sum(c.get("SQ_BUSY_CYCLES", [1])), 1)
        line = (f"{k:44s} dispatches {n:3d}  VALU wave-insts/dispatch {valu:12.0f}  (x4 cyc / 1024 SIMDs / 2.4 GHz = {valu * 4 / 1024 / 2.4e3:7.1f} us)  "
                f"HBM bytes/dispatch {hbm / 1e6:8.1f} MB")
        print(line)
        fo.write(line + "\n")
PY
