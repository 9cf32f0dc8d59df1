#!/bin/bash
# cartpole after the stage-trig rotation: the whole GPU suite, then the cartpole numbers
python -m pytest tests -m gpu -x -q > gpurun_out/r3_qcp_tests.log 2>&1; tail -3 gpurun_out/r3_qcp_tests.log
for cfg in "--env qcp-su --envs 65536 --live-dr 7" "--env qcp-su --envs 65536" "--env qcp-su --envs 32768 --live-dr 7" "--env qcp-su --envs 4096" "--env qcp-su --envs 131072"; do
for var in "" ws ws64 plain; do
VS_ROLLOUT_VARIANT=$var python bench.py --no-cpu-baseline --no-extras --steps 300 --warmup 30 $cfg 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); r = d['roofline']
print('%-45s %-6s | %.3e | %s %.4f ms per %d steps | frac %.3f' % ('$cfg', '${var:-auto}', d['value'], r['kernel'], r['kernel_ms'], d['config']['chunk'], r['frac']))"
done; done
