#!/bin/bash
# every family, records on/off, plain kernel vs wave-specialised kernel (where it applies)
for e in omo bob qq-su qcp-su qbb qq-st qcp-st pend bob-d; do for r in 1 0; do for var in plain ws; do
  VS_ROLLOUT_VARIANT=$var timeout -k 10 200 python bench.py --env $e --record $r --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-7s rec=$r %-5s | %.3e env-steps/s | kernel %.4f ms per 100 steps | %.0f GB/s' % ('$e', '$var', d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"
done; done; done
