cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 400 --warmup 20 --env AntPush-v0 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-60s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
run --envs 2048
run --envs 2048 --lanes 16
run --envs 2048 --lanes 64
run --envs 4096
run --envs 4096 --lanes 16
run --envs 2048 --opt ls_fast_iterations=5
run --envs 2048 --wpb 2
run --envs 2048 --opt solver_tolerance=1e-5
run --envs 2048 --opt solver_rtol=1e-5
