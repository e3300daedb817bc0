# A/B: experiment builds (csrc/exp_ant_<name>.so, NB = 0 + 1 instantiations) on AntPush-v0 / 2048 envs and on the default bench
cd $GRAFT_REPO_ROOT
for f in mujoco_maze_amd/csrc/exp_ant_*.so; do
  for rep in 1 2; do
  MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/$f python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc --env AntPush-v0 --envs 2048 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f  AntPush %.3f M env-steps/s  kernel %.4f ms  bad %d' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
  done
  MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/$f python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f  AntUMaze %.3f M env-steps/s  kernel %.4f ms  bad %d' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done | tee gpurun_out/push_variants.txt
