#!/bin/bash
# A/B of a library variant on the large-batch step (16.7 M envs, lean): default | variant, three rounds on one box
run() { python bench.py --no-cpu-baseline --no-extras --mode step --envs 16777216 --steps 30 --warmup 5 --lean-step 1 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r = d['roofline']; print('$1: kernel %.4f ms  %.0f GB/s  frac %.3f' % (r['kernel_ms'], r['achieved'], r['frac']))"; }
for rep in 1 2 3; do
  unset VS_LIB_PATH; run default
  export VS_LIB_PATH=$PWD/scratch/r3/lib_$1.so; run $1
done
