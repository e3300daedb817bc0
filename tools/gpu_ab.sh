# A/B on the GPU box: every csrc/exp_ant_*.so against the product library, AntUMaze-v0 at the batch sizes in $ENVS (default 4096 8192 16384)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for lib in product $(ls mujoco_maze_amd/csrc/exp_ant_*.so 2>/dev/null); do
  for n in ${ENVS:-4096 8192 16384}; do
    if [ $lib = product ]; then unset MZ_LIBMAZESTEP_EXPERIMENT; else export MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/$lib; fi
    python bench.py --steps 500 --warmup 20 --envs $n --env ${ENVID:-AntUMaze-v0} --no-cpu-baseline --no-live-pmc --sustained 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-45s %6d envs  %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$lib', $n, d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
  done
done | tee gpurun_out/ab/ant_ab.txt
