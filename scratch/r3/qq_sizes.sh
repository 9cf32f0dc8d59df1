#!/bin/bash
# qq_sizes.sh <libs...>: QQube swing-up / stabilisation, automatic kernel, 1 024 .. 65 536 envs, shipped library against variant builds
for pass in 1 2; do for lib in "" "$@"; do
  if [ -n "$lib" ]; then export VS_LIB_PATH=$PWD/scratch/r3/lib_$lib.so; else unset VS_LIB_PATH; fi
  for env in qq-su qq-st; do for n in 1024 4096 16384 32768 65536; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $env --envs $n --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-8s %-6s %6d | %.3e | kernel %.4f ms | %-16s' % ('${lib:-shipped}', '$env', $n, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1
done; done; done; done
