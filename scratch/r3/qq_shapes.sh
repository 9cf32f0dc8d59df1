#!/bin/bash
# QQube at 65 536 envs: the shapes against each other (and under a live randomizer), two passes
for pass in 1 2; do for dr in 0 7; do for v in g256 g64 ws ws64; do
  VS_ROLLOUT_VARIANT=$v timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env qq-su --envs 65536 --live-dr $dr --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('qq-su live-dr $dr %-5s | %.3e | kernel %.4f ms | %-16s' % ('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1
done; done; done
