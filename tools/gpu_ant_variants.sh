# A/B: experiment builds of the plain ant's kernel (csrc/exp_ant_<name>.so) on the default bench
cd $GRAFT_REPO_ROOT
for f in mujoco_maze_amd/csrc/exp_ant_*.so; do
  for rep in 1 2; do
  MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/$f python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$f  %.3f M env-steps/s  kernel %.4f ms  bad %d' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
  done
done | tee gpurun_out/ant_variants.txt
