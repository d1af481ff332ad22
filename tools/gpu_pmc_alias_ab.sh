#!/bin/bash
# read traffic of the x2 tile kernel with and without TILE_LDS_ALIAS (four vs three workgroups per CU): request-size counters only
set -u
export TMPDIR=/tmp
for v in alias noalias; do
  lib=multi_frame_super_resolution_amd/lib/libmfsr_hip.so; [ $v = noalias ] && lib=build_ab/libmfsr_noalias.so
  out=gpurun_out/${1:-r04pmc_alias}/$v; mkdir -p $out
  MFSR_LIB=$PWD/$lib timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum \
     --kernel-include-regex "accumulate2xTile" --output-format csv -d $out/p4 -- python3 bench.py --no-cpu-baseline --no-e2e --no-isolated --steps 1 --warmup 0 > $out/p4.log 2>&1
  python3 - $out $v <<'PY'
import csv,glob,sys,collections
per=collections.defaultdict(list)
for f in glob.glob(sys.argv[1]+"/p4/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        per[(r["Dispatch_Id"],r["Counter_Name"])].append(float(r["Counter_Value"]))
d=collections.defaultdict(dict)
for (i,c),v in per.items(): d[i][c]=sum(v)
rows=sorted(d.items(), key=lambda kv:int(kv[0]))
for i,c in rows:
    b=32*c.get("TCC_EA0_RDREQ_32B_sum",0)+64*c.get("TCC_EA0_RDREQ_64B_sum",0)+128*c.get("TCC_EA0_RDREQ_128B_sum",0)
    print(sys.argv[2], "dispatch", i, f"32B {c.get('TCC_EA0_RDREQ_32B_sum',0):.4g} 64B {c.get('TCC_EA0_RDREQ_64B_sum',0):.4g} 128B {c.get('TCC_EA0_RDREQ_128B_sum',0):.4g} all {c.get('TCC_EA0_RDREQ_sum',0):.4g} -> {b/1e9:.3f} GB read")
PY
done
