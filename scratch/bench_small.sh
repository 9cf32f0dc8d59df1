#!/bin/bash
for e in qbb omo qq-su; do for envs in 32768 16384; do for var in plain ws; do
  VS_ROLLOUT_VARIANT=$var timeout -k 10 200 python bench.py --env $e --envs $envs --record 1 --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-7s envs $envs %-5s | %.3e env-steps/s | kernel %.4f ms per 100 steps' % ('$e', '$var', d['value'], d['roofline']['kernel_ms']))"
done; done; done
