#!/bin/bash
# GPU box helper: the chain kernels (Swimmer / Reacher / user chains) after a change — their GPU parity tests, bench lines, phase timers
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -k "swimmer or reacher or chain or Swimmer or Reacher or mjcf" 2>&1 | tail -4
for e in SwimmerUMaze-v0 ReacherUMaze-v0 SwimmerPush-v0; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --env $e 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done
[ -f mujoco_maze_amd/csrc/exp_SWPROF.so ] && { python tools/exp_swimmer_prof.py SwimmerUMaze-v0; python tools/exp_swimmer_prof.py ReacherUMaze-v0; }
