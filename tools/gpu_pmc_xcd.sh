#!/bin/bash
# read traffic + burst time of the x2 tile kernel with the XCD-aware tile order (MFSR_XCD_REMAP=1) against launch order
set -u
export TMPDIR=/tmp
for v in 0 1; do
  out=gpurun_out/${1:-r04pmc_xcd}/remap$v; mkdir -p $out
  MFSR_XCD_REMAP=$v timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum \
     --kernel-include-regex "accumulate2xTile" --output-format csv -d $out/p4 -- python3 bench.py --no-cpu-baseline --no-e2e --no-isolated --steps 1 --warmup 0 > $out/p4.log 2>&1
  python3 - $out remap$v <<'PY'
import csv,glob,sys,collections
per=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/p4/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"],r["Counter_Name"])].append(float(r["Counter_Value"]))
d=collections.defaultdict(dict)
for (i,c),v in per.items(): d[i][c]=sum(v)
for i,c in sorted(d.items(), key=lambda kv:int(kv[0])):
    b=32*c.get("TCC_EA0_RDREQ_32B_sum",0)+64*c.get("TCC_EA0_RDREQ_64B_sum",0)+128*c.get("TCC_EA0_RDREQ_128B_sum",0)
    print(sys.argv[2], "dispatch", i, f"-> {b/1e9:.3f} GB read")
PY
  for r in 1 2 3; do MFSR_XCD_REMAP=$v python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('remap$v', d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['isolated']['avg_launch_ms'])"; done
done
