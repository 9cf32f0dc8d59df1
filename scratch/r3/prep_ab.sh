#!/bin/bash
# prep_ab.sh: ball-on-beam (variant bp) and pendulum (variant pp) with the action pre-processing on the generator wave, against the main build
one() { timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --steps 300 --warmup 30 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-8s %-6s %6d | %.3e | kernel %.4f ms | %-16s' % ('$3', '$1', $2, d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))" || exit 1; }
for pass in 1 2; do for n in 4096 32768 65536; do
  unset VS_LIB_PATH; one bob $n main; one bob-d $n main; one pend $n main
  export VS_LIB_PATH=$PWD/scratch/r3/lib_bp.so; one bob $n bp
  export VS_LIB_PATH=$PWD/scratch/r3/lib_pp.so; one pend $n pp
done; done
