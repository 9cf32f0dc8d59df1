#!/bin/bash
# A/B of the large-batch step kernel on one box: {k_step_v4, k_step} x {lean, default}, three rounds
run() { python bench.py --no-cpu-baseline --no-extras --mode step --envs 16777216 --steps 30 --warmup 5 --lean-step $1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r = d['roofline']; print('$2 lean $1: kernel %.4f ms  %.0f GB/s  frac %.3f' % (r['kernel_ms'], r['achieved'], r['frac']))"; }
for rep in 1 2 3; do
  run 1 v4; run 0 v4
  VS_NO_STEP_V4=1 run 1 scalar; VS_NO_STEP_V4=1 run 0 scalar
done
