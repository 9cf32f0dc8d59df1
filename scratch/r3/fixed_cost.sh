#!/bin/bash
# fixed cost per launch: the cartpole at 65 536 envs in 64-env workgroups, with and without the live randomizer, 4 .. 400 steps per launch
[ -n "$1" ] && export VS_LIB_PATH=$PWD/scratch/r3/lib_$1.so
for dr in 0 7; do for chunk in 4 8 16 32 100 400; do
  VS_ROLLOUT_VARIANT=ws64 python bench.py --no-cpu-baseline --no-extras --env qcp-su --envs 65536 --live-dr $dr --chunk $chunk --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('live-dr $dr chunk %4d | kernel %8.2f us | %.3f us per step' % ($chunk, d['roofline']['kernel_ms'] * 1e3, d['roofline']['kernel_ms'] * 1e3 / $chunk))"
done; done
