#!/bin/bash
# A/B of library builds (same C-ABI) on one GPU box: AB_LIBS="path1 path2" [AB_ENV="NAME=v"] ; two interleaved rounds
for round in 1 2; do
  for l in ${AB_LIBS}; do
    env MFSR_LIB=$PWD/$l ${AB_ENV:-X=1} python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$l ${AB_ENV:-}', d['ms_per_step'], r['avg_launch_ms'], r['frames_per_launch'], r['frac'])"
  done
done
