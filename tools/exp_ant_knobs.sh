# GPU box (experiment): launch knobs of the plain ant at the metric's batch (lanes per env, waves per workgroup, waves per SIMD)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-64s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
run --envs 4096
run --envs 4096 --wpb 2
run --envs 4096 --wpb 4
run --envs 4096 --lanes 32
run --envs 4096 --opt waves_per_simd=2
run --envs 8192 --wpb 2
run --envs 8192 --opt waves_per_simd=1
run --env AntPush-v0 --envs 2048 --wpb 2
run --env AntPush-v0 --envs 4096 --wpb 2
run --env AntPush-v0 --envs 4096
run --env SwimmerUMaze-v0 --envs 4096
run --env SwimmerUMaze-v0 --envs 16384
