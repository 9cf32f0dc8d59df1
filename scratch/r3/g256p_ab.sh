#!/bin/bash
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $1 --envs 65536 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-8s %-6s | %.3e | kernel %.4f ms | %-16s' % ('$2', '$1', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for pass in 1 2 3; do for env in qq-su omo pend; do
  unset VS_LIB_PATH; one $env main
  export VS_LIB_PATH=$PWD/scratch/r3/lib_g256p.so; one $env g256p
done; done
