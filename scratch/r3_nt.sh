#!/bin/bash
# plain against non-temporal record stores (diagnostic build -DVS_REC_NT -> scratch/libvecsim_nt.so), bench default and others
for lib in "" scratch/libvecsim_nt.so; do
  for cfg in "--env qq-su --envs 65536 --record 1" "--env qq-su --envs 65536 --record 2" "--env qq-su --envs 131072 --record 1 --chunk 100" "--env omo --envs 65536 --record 1" "--env bob --envs 65536 --record 1" "--env qq-su --envs 1048576 --record 1 --chunk 20"; do
    VS_LIB_PATH=$lib python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('%-26s %-60s | %.3e | kernel %.4f ms | %s | frac %.3f' % ('${lib:-default}', '$cfg', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['roofline']['frac']))"
  done
done
