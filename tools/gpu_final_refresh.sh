#!/bin/bash
# End-of-round refresh of the judged artefacts (runs on the GPU box through gpurun):
#   1. PMC passes of the warp+fuse launches (x2 workloads) -> gpurun_out/<tag>/pmc/{summary.txt,fuse_traffic.json}
#   2. rocprofv3 --kernel-trace --stats of the default bench command -> <tag>/trace + bench_profiled.json
#   3. the default bench line (with cpu_baseline and end_to_end) -> <tag>/bench_n1.json; the other workloads' lines
# Usage: tools/gpu_final_refresh.sh <tag>
set -u
tag=${1:-final}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
tools/gpu_pmc_workloads.sh $tag/pmc 4k16_rggb_x2 8k8_rggb_x2 1080p5_gray_x2 > "$out/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.log"; }
echo "pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_profiled.json" 2> "$out/bench_profiled.err" || echo "trace run failed"
echo "trace done"
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" || echo "bench failed"
echo "bench done"
for wl in 1080p5_gray_x2 4k16_rggb_x4 8k8_rggb_x2; do
  python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 > "$out/bench_$wl.json" 2>/dev/null || echo "$wl failed"
done
echo "workloads done"
find "$out/trace" -name "*kernel_stats.csv" | head -2
