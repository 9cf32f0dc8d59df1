#!/bin/bash
for e in qq-su qcp-su bob; do for k in 0 1 3 7; do for v in plain ws; do
  VS_ROLLOUT_VARIANT=$v python bench.py --no-cpu-baseline --env $e --live-dr $k --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-7s live-dr $k %-5s | %.3e | kernel %.4f ms | %s | mean len %.1f nan %d' % ('$e', '$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['episodes']['mean_length'], d['nan_flags']))"
done; done; done
