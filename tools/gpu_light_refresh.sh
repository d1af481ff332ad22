#!/bin/bash
# The bench lines and kernel-stats summaries of tools/gpu_final_refresh.sh without its PMC passes and probes (for a HEAD whose
# accumulate_fast.hip / accumulate_common.hpp -- the stamp of profiles/fuse_traffic.json -- did not change):
#   tools/gpu_light_refresh.sh <tag>   -> gpurun_out/<tag>/{bench_n1.json, bench_profiled.json, bench_timed_region.json, trace/, trace_timed/}
set -u
tag=${1:-light}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py > "$out/bench_profiled.json" 2> "$out/bench_profiled.err" || echo "trace run failed"
python3 bench.py > "$out/bench_n1.json" 2> "$out/bench_n1.err" || echo "bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace_timed" -- python3 bench.py --no-e2e --no-isolated --no-cpu-baseline > "$out/bench_timed_region.json" 2> /dev/null || echo "timed-region trace failed"
echo "light refresh done"
