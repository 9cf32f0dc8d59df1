#!/bin/bash
# throughput of the fused kernel vs occupancy (envs per GPU), plain / ws, record on / off
for envs in 65536 131072 262144 1048576; do
  for rec in 0 1; do
    for var in plain ws; do
      if [ $envs -gt 131072 ] && [ $var = ws ]; then continue; fi
      chunk=100; [ $envs -ge 1048576 ] && chunk=20
      VS_ROLLOUT_VARIANT=$var python bench.py --no-cpu-baseline --envs $envs --record $rec --chunk $chunk --steps 50 --warmup 5 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('envs %8d rec %d %-5s | %.3e env-steps/s | kernel %.4f ms per %d steps' % ($envs, $rec, '$var', d['value'], d['roofline']['kernel_ms'], d['config']['chunk']))"
    done
  done
done
