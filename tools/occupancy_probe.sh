#!/bin/bash
# GPU box helper: kernel time vs env count for a lane width (detects multi-round residency)
for g in 16 32; do for n in 512 1024 2048 4096 8192; do
  python $(dirname $0)/../bench.py --steps 60 --warmup 20 --lanes $g --envs $n --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes', d['config']['lanes_per_env'], 'envs', d['config']['envs_per_gpu'], 'kernel_ms %.3f'%d['roofline']['kernel_ms'], 'env-steps/s %.0f'%d['value'])"
done; done
