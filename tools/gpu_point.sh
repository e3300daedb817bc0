#!/bin/bash
# GPU box helper: the Point kernels after a change — their GPU tests (golden detector moves included), bench lines, phase timers
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "point or Point or golden or billiard or Billiard" 2>&1 | tail -3
for e in PointUMaze-v0 Point4Rooms-v0 PointPush-v0 PointBilliard-v0 PointFall-v0; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --env $e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done
[ -f mujoco_maze_amd/csrc/exp_PROF.so ] && python tools/exp_point_prof.py PointUMaze-v0
