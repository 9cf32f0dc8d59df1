#!/bin/bash
one() { timeout -k 10 180 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-7s %8d %-6s | %.3e | kernel %.4f ms | %-16s' % ('$1', $2, '$3', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for env in qq-su qq-st; do for n in 73728 86016 90112 94208 98304; do
  unset VS_ROLLOUT_VARIANT; one $env $n auto
  export VS_ROLLOUT_VARIANT=plain; one $env $n plain
  export VS_ROLLOUT_VARIANT=g64; one $env $n g64
done; done
