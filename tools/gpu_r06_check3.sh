#!/bin/bash
# GPU box: GPU suite + bench lines of the BASELINE workloads + the Fall-family outlier statistics (round 6 check after the solver clean-up)
cd $GRAFT_REPO_ROOT; out=gpurun_out/check3_r06; mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -8 > $out/gpu_suite.log; cat $out/gpu_suite.log
for w in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "AntUMaze-v0 8192"; do set -- $w
  python bench.py --no-cpu-baseline --no-live-pmc --steps 500 --warmup 20 --sustained 0 --env $1 --envs $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f M kernel %.4f ms %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['launch']))"
done | tee $out/bench_quick.txt
MZ_PS_CONFIGS="AntMultiFall-v0,AntFall-v0,AntMultiFall-v0@8,AntFall-v0@2" python tools/parity_stats.py long 2>/dev/null > $out/parity_fall.md; cat $out/parity_fall.md
