#!/bin/bash
# the cartpole, two-role against three-role 64-env workgroups (and the automatic choice), 1 024 .. 65 536 envs, with and without the randomizer
for n in 1024 4096 16384 32768 65536; do for dr in 0 7; do for v in ws64 g64 auto; do
  if [ $v = auto ]; then unset VS_ROLLOUT_VARIANT; else export VS_ROLLOUT_VARIANT=$v; fi
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env qcp-su --envs $n --live-dr $dr --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('qcp-su %6d live-dr $dr %-5s | %.3e | kernel %.4f ms | %-16s' % ($n, '$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1
done; done; done
