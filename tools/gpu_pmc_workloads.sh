#!/bin/bash
# PMC passes of the warp+fuse launches for every bench workload (runs on the GPU box through gpurun):
#   pass 1: SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES   pass 2: FETCH_SIZE   pass 3: WRITE_SIZE
#   pass 4: TCC_EA0_RDREQ_{32B,64B,128B}_sum + TCC_EA0_RDREQ_sum (the read requests by size: exact bytes at the L2's memory side, so the
#           guide's x2 correction of FETCH_SIZE is applied only where the requests ARE 128-byte ones)   pass 5: TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
# (separate passes, kernel-trace only, as MI355X_MICROARCH.md prescribes), restricted to the accumulate kernels.
# Usage: tools/gpu_pmc_workloads.sh <tag> [workload ...]   -> gpurun_out/<tag>/<workload>/p{1,2,3}, summary by tools/pmc_fuse_summary.py
set -u
tag=${1:-r02pmc}
shift || true
wls=${@:-4k16_rggb_x2 4k16_rggb_x4 8k8_rggb_x2 1080p5_gray_x2}
export TMPDIR=/tmp
for wl in $wls; do
  out=gpurun_out/$tag/$wl
  mkdir -p "$out"
  B="python3 bench.py --workload $wl --no-cpu-baseline --no-e2e --steps 1 --warmup 0"
  i=0
  for grp in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --pmc $grp --kernel-include-regex "${KERNEL_RE:-accumulate}" --output-format csv -d "$out/p$i" -- $B > "$out/p$i.log" 2>&1 \
      || { echo "$wl pass $i failed"; tail -3 "$out/p$i.log"; exit 1; }
    echo "$wl pass $i done"
  done
done
python3 tools/pmc_fuse_summary.py gpurun_out/$tag $wls
