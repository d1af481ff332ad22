#!/bin/bash
# A/B of one environment knob on one GPU box (interleaved, two rounds): AB_VAR=NAME AB_VALUES="a b" [BENCH_ARGS=...]
for round in 1 2; do
  for v in ${AB_VALUES}; do
    env "${AB_VAR}=$v" python bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${AB_VAR}=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
