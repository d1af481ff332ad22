#!/bin/bash
# A/B of accumulate-kernel builds on one GPU box (interleaved, two rounds)
for round in 1 2; do
  for v in ${AB_VARIANTS:-t4 n3 n4}; do
    MFSR_LIB=$PWD/multi_frame_super_resolution_amd/lib/ab/libmfsr_$v.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  done
done
