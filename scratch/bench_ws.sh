#!/bin/bash
# the two fused kernels on the headline configuration, records on / off
for rec in 1 0; do for var in plain ws; do
  VS_ROLLOUT_VARIANT=$var python bench.py --no-cpu-baseline --record $rec --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-5s rec $rec | %.3e | kernel %.4f ms | %s' % ('$var', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))"
done; done
