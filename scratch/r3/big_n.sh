#!/bin/bash
# big_n.sh: every family at 131 072 envs (QQube also 98 305 .. 1 M): the automatic choice (sub-launched three-role kernel) against k_rollout
one() { timeout -k 10 180 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --chunk $4 --steps 100 --warmup 10 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-7s %8d %-6s chunk %3d | %.3e | kernel %.4f ms | %-16s' % ('$1', $2, '$3', $4, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for env in qq-su omo bob pend qcp-su qbb qq-st bob-d; do
  unset VS_ROLLOUT_VARIANT; one $env 131072 auto 400
  export VS_ROLLOUT_VARIANT=plain; one $env 131072 plain 400
done
for n in 98304 114688 196608 262144; do
  unset VS_ROLLOUT_VARIANT; one qq-su $n auto 400
  export VS_ROLLOUT_VARIANT=plain; one qq-su $n plain 400
done
unset VS_ROLLOUT_VARIANT; one qq-su 1048576 auto 20
export VS_ROLLOUT_VARIANT=plain; one qq-su 1048576 plain 20
