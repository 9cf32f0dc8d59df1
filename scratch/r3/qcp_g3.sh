#!/bin/bash
# qcp_g3.sh <libs...>: the cartpole at 65 536 envs with / without the live randomizer, three-role 64-env shape (and the two-role one)
for pass in 1 2; do for lib in "$@"; do export VS_LIB_PATH=$PWD/scratch/r3/lib_$lib.so; for dr in 7 0; do for v in g64 ws64; do
  VS_ROLLOUT_VARIANT=$v timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env qcp-su --envs 65536 --live-dr $dr --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$lib qcp-su live-dr $dr %-5s | %.3e | kernel %.4f ms | %-16s' % ('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1
done; done; done; done
