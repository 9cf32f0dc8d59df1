#!/bin/bash
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-8s %-6s %6d | %.3e | kernel %.4f ms | %-16s' % ('$3', '$1', $2, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for pass in 1 2; do for n in 4096 32768; do
  unset VS_LIB_PATH; one omo $n main
  export VS_LIB_PATH=$PWD/scratch/r3/lib_op.so; one omo $n op
done; done
