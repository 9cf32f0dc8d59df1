#!/bin/bash
# usage: bench_variants.sh  -> prints value / kernel_ms / frac for a set of bench configs
for m in "--record 1" "--record 0" "--record 1 --per-env-params 0" "--mode step" "--record 1 --envs 1048576 --chunk 20 --steps 200 --warmup 20" "--mode step --envs 16777216 --steps 50 --warmup 5"; do
  timeout -k 10 200 python bench.py $m --no-cpu-baseline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['config']['workload'][6:24], d['config']['workload'][-62:], '| %.3e env-steps/s | kernel %.4f ms | %.0f GB/s | frac %.3f' % (d['value'], d['roofline']['kernel_ms'], d['roofline']['achieved'], d['roofline']['frac']))"
done
