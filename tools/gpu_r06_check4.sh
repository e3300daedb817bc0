#!/bin/bash
# GPU box: GPU suite + bench lines of the BASELINE workloads + the block / ball / fall mazes (round 6: per-env goals, Hessian entry order)
cd $GRAFT_REPO_ROOT; out=gpurun_out/check4_r06; mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -12 > $out/gpu_suite.log; cat $out/gpu_suite.log
for w in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "AntUMaze-v0 8192"; do set -- $w
  python bench.py --no-cpu-baseline --no-live-pmc --steps 500 --warmup 20 --sustained 0 --env $1 --envs $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 %.3f M kernel %.4f ms %s' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['launch']))"
done | tee $out/bench_quick.txt
bash tools/gpu_other_configs.sh; cp gpurun_out/other_configs.txt $out/
