#!/bin/bash
# live domain randomisation (DomainRandWrapperLive on the device) across the rollout kernel variants
for spec in "qq-su 65536 7" "qq-su 65536 0" "qcp-su 65536 7" "qcp-su 65536 0" "qq-su 4096 7" "omo 65536 3" "pend 65536 3" "qq-st 65536 7"; do
  set -- $spec
  for v in plain ws ws64 g64 g256 auto; do
    if [ $v = auto ]; then unset VS_ROLLOUT_VARIANT; else export VS_ROLLOUT_VARIANT=$v; fi
    python bench.py --no-cpu-baseline --no-extras --env $1 --envs $2 --live-dr $3 --chunk 100 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('%-7s %6d live-dr $3 %-5s | %.3e | kernel %.4f ms | %-18s | mean len %.1f nan %d' % ('$1', $2, '$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel'], d['episodes']['mean_length'], d['nan_flags']))"
  done
done
# the two-role 64-env shape between 256 and 384 envs per compute unit (needs three waves per SIMD)
for v in plain ws64 auto; do
  if [ $v = auto ]; then unset VS_ROLLOUT_VARIANT; else export VS_ROLLOUT_VARIANT=$v; fi
  python bench.py --no-cpu-baseline --no-extras --env qq-su --envs 98304 --chunk 100 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('qq-su 98304 %-5s | %.3e | kernel %.4f ms | %-18s' % ('$v', d['value'], d['roofline']['kernel_ms'], d['roofline']['kernel']))"
done
