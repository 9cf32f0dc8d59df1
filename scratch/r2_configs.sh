#!/bin/bash
# per-GPU rank shards of BASELINE.json's configs 2-5 through bench.py (fused mode, records on, HIP-event kernel time)
run() {
  python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r = d['roofline']
print('%-50s | %.3e env-steps/s per GPU | %s %.4f ms per %d steps | %.0f GB/s alg = %.3f of 8 TB/s' % ('$*', d['value'], r['kernel'], r['kernel_ms'], d['config']['chunk'], r['achieved'], r['frac']))"
}
run --env qq-su --envs 4096
run --env qq-su --envs 4096 --record 0
run --env qcp-su --envs 65536 --live-dr 7
run --env qbb --envs 32768
run --env qq-su --envs 65536 --record 2
run --env qq-su --envs 131072
run --env qq-su --envs 1048576 --chunk 20
python scratch/bench_mixed.py 130560 2>/dev/null | grep -v amdgpu
