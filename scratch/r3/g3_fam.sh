#!/bin/bash
# g3_fam.sh: ball-on-beam and ball balancer, two-role shapes against three-role ones (variant builds bg3 / qbg3)
one() { # env n variant
  timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$1 %6d %-5s | %.3e | kernel %.4f ms | %-16s' % ($2, '$3', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1
}
for pass in 1 2; do
export VS_LIB_PATH=$PWD/scratch/r3/lib_bg3.so
for n in 4096 65536; do for v in ws64 ws g64 g256; do VS_ROLLOUT_VARIANT=$v one bob $n $v; done; done
export VS_LIB_PATH=$PWD/scratch/r3/lib_qbg3.so
for n in 4096 32768 65536; do for v in ws64 ws g64 g256; do VS_ROLLOUT_VARIANT=$v one qbb $n $v; done; done
done
