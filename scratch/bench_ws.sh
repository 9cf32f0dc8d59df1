#!/bin/bash
for rec in 1 0; do
  VS_ROLLOUT_VARIANT=plain python bench.py --no-cpu-baseline --record $rec --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('plain    rec $rec | %.3e | kernel %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
  for r in 1 2 4; do
  VS_ROLLOUT_VARIANT=ws python bench.py --no-cpu-baseline --record $rec --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ws R=$r   rec $rec | %.3e | kernel %.4f ms' % (d['value'], d['roofline']['kernel_ms']))"
  done
done
