#!/bin/bash
# where does k_rollout_ws stop paying?  envs per GPU between 65 536 and 131 072
for envs in 73728 81920 98304 114688; do for rec in 1 0; do for var in plain ws; do
  VS_ROLLOUT_VARIANT=$var python bench.py --no-cpu-baseline --envs $envs --record $rec --steps 60 --warmup 6 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('envs %7d rec $rec %-5s | %.3e | kernel %.4f ms' % ($envs, '$var', d['value'], d['roofline']['kernel_ms']))"
done; done; done
