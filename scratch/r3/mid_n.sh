#!/bin/bash
# mid_n.sh: 256 .. 384 envs per compute unit (65 537 .. 98 304 envs): the automatic choice (two-role 64-env workgroups where E::WS_MID) against k_rollout
one() { timeout -k 10 180 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-7s %8d %-6s | %.3e | kernel %.4f ms | %-16s' % ('$1', $2, '$3', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for env in qq-su omo bob pend qq-st bob-d; do for n in 81920 98304; do
  unset VS_ROLLOUT_VARIANT; one $env $n auto
  export VS_ROLLOUT_VARIANT=plain; one $env $n plain
  export VS_ROLLOUT_VARIANT=g64; one $env $n g64
done; done
