#!/bin/bash
# ab_qq.sh <lib>: QQube 65 536 (headline shape) and 4 096, main build against a variant, three passes
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env qq-su --envs $1 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-8s %6d | %.3e | kernel %.4f ms | %-16s' % ('$2', $1, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for pass in 1 2 3; do for n in 65536 4096; do
  unset VS_LIB_PATH; one $n main
  export VS_LIB_PATH=$PWD/scratch/r3/lib_$1.so; one $n $1
done; done
