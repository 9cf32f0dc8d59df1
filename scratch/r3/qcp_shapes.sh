#!/bin/bash
# qcp_shapes.sh [lib]: the cartpole at 65 536 envs, {live DR 7, none} x {256-env, 64-env workgroups}, 400 and 100 steps per launch
[ -n "$1" ] && export VS_LIB_PATH=$PWD/scratch/r3/lib_$1.so
echo "== lib: ${VS_LIB_PATH:-default}"
for chunk in 400 100; do for dr in 0 7; do for v in ws ws64; do
  VS_ROLLOUT_VARIANT=$v python bench.py --no-cpu-baseline --no-extras --env qcp-su --envs 65536 --live-dr $dr --chunk $chunk --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('qcp-su live-dr $dr chunk $chunk %-5s | %.3e | kernel %.4f ms | %-16s | mean len %.0f episodes %d' % ('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['episodes']['mean_length'], d['episodes']['completed']))"
done; done; done
