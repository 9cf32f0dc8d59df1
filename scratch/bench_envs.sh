#!/bin/bash
for e in omo bob qq-su qcp-su qbb; do for r in 1 0; do
  timeout -k 10 200 python bench.py --env $e --record $r --steps 400 --warmup 100 --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e rec=$r | %.3e env-steps/s | kernel %.4f ms per 100 steps | %.0f GB/s' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"
done; done
