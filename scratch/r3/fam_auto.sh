#!/bin/bash
# the automatic kernel of every family at 4 096 / 32 768 / 65 536 envs (per-env constants, auto-reset, record mode 1)
for env in omo bob bob-d qq-su qq-st qcp-su qbb pend; do for n in 4096 32768 65536; do
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $env --envs $n --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-7s %6d | %.3e | kernel %.4f ms | %-16s | frac %.3f' % ('$env', $n, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['roofline']['frac']))" || exit 1
done; done
