# GPU box (experiment): the one-block ant with 32 contact slots (csrc/exp_ant_nc32.so: make dev1 DEVFLAGS=-DMZ_NC1=32) — four 16-lane waves
# per CU fit the LDS then.  Large batches at 16 and 32 lanes per env, and the overflow flags of a soak.
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-64s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
for lib in product nc32; do
  if [ $lib = product ]; then unset MZ_LIBMAZESTEP_EXPERIMENT; else export MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/exp_ant_nc32.so; fi
  echo "== $lib"
  for e in AntPush-v0 AntFall-v0; do for n in 2048 4096 8192; do for l in 16 32; do run --env $e --envs $n --lanes $l; done; done; done
done
