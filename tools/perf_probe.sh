#!/bin/bash
# GPU box helper: short bench lines (default and other lane widths) + in-kernel phase timers -> stdout
cd $GRAFT_REPO_ROOT
for L in ${LANES:-0}; do
  python bench.py --steps ${STEPS:-300} --warmup 10 --no-cpu-baseline --lanes $L 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lanes', '$L', 'value %.3f M/s' % (d['value']/1e6), 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
python tools/phase_profile.py ${PLANES:-32} 2>/dev/null
