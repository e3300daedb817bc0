#!/bin/bash
# GPU box helper: sweep lanes-per-env and print one bench line each
for g in 8 16 32 64; do
  python bench.py --steps 100 --warmup 30 --lanes $g --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', d['config']['lanes_per_env'], 'env-steps/s %.0f'%d['value'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'])"
done
