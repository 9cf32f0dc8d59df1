#!/bin/bash
# which randomised parameter makes live DR expensive for the cartpole (rail_length is the 4th: it moves the state box)
for k in 0 1 3 4 5 7; do
  for v in ws ws64; do
    export VS_ROLLOUT_VARIANT=$v
    python bench.py --no-cpu-baseline --no-extras --env qcp-su --envs 65536 --live-dr $k --chunk 100 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('qcp-su live-dr $k %-5s | %.3e | kernel %.4f ms | %-18s | mean len %.1f episodes %d nan %d' % ('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['episodes']['mean_length'], d['episodes']['completed'], d['nan_flags']))"
  done
done
