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
tools/gpu_pmc_workloads.sh $tag/pmc 4k16_rggb_x2 4k16_rggb_x4 8k8_rggb_x2 1080p5_gray_x2 > "$out/pmc.log" 2>&1 || { echo "pmc failed"; tail -5 "$out/pmc.log"; }
echo "pmc done"
cp "$out/pmc/fuse_traffic.json" profiles/fuse_traffic.json   # the bench lines below report against the fresh counters
tools/gpu_pmc_align.sh $tag/pmc_align > "$out/pmc_align.log" 2>&1 || echo "align pmc failed"
echo "align pmc done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_profiled.json" 2> "$out/bench_profiled.err" || echo "trace run failed"
echo "trace done"
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" || echo "bench failed"
echo "bench done"
for wl in 1080p5_gray_x2 4k16_rggb_x4 8k8_rggb_x2 8k64_rggb_x2; do
  python3 bench.py --workload $wl --no-cpu-baseline --steps 10 --warmup 3 > "$out/bench_$wl.json" 2>/dev/null || echo "$wl failed"
done
echo "workloads done"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_timed" -- python3 bench.py --no-e2e --no-isolated --no-cpu-baseline > "$out/bench_timed_region.json" 2> /dev/null || echo "timed-region trace failed"
python3 bench.py --no-async-fuse --no-cpu-baseline --no-e2e > "$out/bench_no_async_fuse.json" 2>/dev/null || echo "no-async run failed"
python3 tools/pcie_duplex.py > "$out/pcie_duplex.json" 2>/dev/null || echo "pcie probe failed"
rocprofv3 --kernel-trace --output-format csv -d "$out/rccl" -- python3 tools/rccl_under_fuse.py > "$out/rccl_under_fuse.json" 2> /dev/null || echo "rccl probe failed"
python3 tools/rccl_trace_summary.py "$out/rccl" > "$out/rccl_under_fuse_trace.txt" 2>&1
echo "probes done"
find "$out/trace" -name "*kernel_stats.csv" | head -2
