#!/bin/bash
# GPU box helper: sweep lanes-per-env x waves-per-workgroup and print one bench line each
for g in 8 16 32 64; do for w in 1 2 4; do
  python $(dirname $0)/../bench.py --steps 100 --warmup 30 --lanes $g --wpb $w --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', d['config']['lanes_per_env'], 'wpb', d['config']['waves_per_block'], 'env-steps/s %.0f'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'bad', d['config']['bad_envs'])" || echo "lanes $g wpb $w failed"
done; done
