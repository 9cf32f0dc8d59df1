#!/bin/bash
# cfg.sh [lib-variant-name]: BASELINE configs 2-5 + headline shapes through bench.py, one line each
[ -n "$1" ] && export VS_LIB_PATH=$PWD/scratch/r3/lib_$1.so
run() {
  python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r = d['roofline']
print('%-46s | %.3e env-steps/s | %s %.4f ms per %d steps | frac %.3f' % ('$*', d['value'], r['kernel'], r['kernel_ms'], d['config']['chunk'], r['frac']))"
}
echo "== lib: ${VS_LIB_PATH:-default}"
run --env qq-su --envs 65536
run --env qq-su --envs 4096
run --env qcp-su --envs 65536 --live-dr 7
run --env qcp-su --envs 65536
run --env qbb --envs 32768
run --env qbb --envs 65536
run --env bob --envs 65536
