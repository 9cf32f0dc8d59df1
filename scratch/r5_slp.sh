#!/bin/bash
# -fslp-vectorize (packed fp32 VALU ops) for the physics-bound families only: qcp (family 3) and qbb (family 4) against the shipped build
run() {
  python bench.py --no-cpu-baseline --no-extras --steps 200 --warmup 20 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r = d['roofline']
print('%-8s %-44s | %.3e env-steps/s | %s %.4f ms per %d steps' % ('$TAG', '$*', d['value'], r['kernel'], r['kernel_ms'], d['config']['chunk']))"
}
for lib in shipped slp; do
  if [ $lib = slp ]; then export VS_LIB_PATH=$PWD/scratch/slp/libvecsim_slp.so; else unset VS_LIB_PATH; fi
  TAG=$lib
  run --env qcp-su --envs 65536
  run --env qcp-su --envs 65536 --live-dr 7
  run --env qbb --envs 32768
  run --env qbb --envs 65536
  run --env qcp-su --envs 262144 --chunk 100
done
