#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + stats, then separate PMC
# passes (the guide forbids mixing --pmc with trace domains other than kernel-trace).
# Usage: tools/gpu_profile.sh <tag>   -> gpurun_out/<tag>/...
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="python3 bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- $B --steps 2 --warmup 1 > "$out/bench_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d "$out/pmc_sq" -- $B --steps 1 --warmup 0 > "$out/bench_pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- $B --steps 1 --warmup 0 > "$out/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- $B --steps 1 --warmup 0 > "$out/bench_pmc_write.log" 2>&1
find "$out" -name "*.csv" | head -20
