#!/bin/bash
# A/B of accumulate-kernel variants on one GPU box (interleaved, two rounds); variants = MFSR_STRIP_TILE values
for round in 1 2; do
  for v in ${AB_VARIANTS:-0 1}; do
    MFSR_STRIP_TILE=$v python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tile=$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
