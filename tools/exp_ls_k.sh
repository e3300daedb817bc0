# GPU box: kernel time against the number of unit-step iterations in front of the exact line search (option ls_fast_iterations)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-70s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
for k in 3 4 5 6 3 5; do run --env AntUMaze-v0 --envs 4096 --opt ls_fast_iterations=$k; done
for k in 3 5; do run --env AntUMaze-v0 --envs 8192 --opt ls_fast_iterations=$k; run --env Ant4Rooms-v0 --envs 4096 --opt ls_fast_iterations=$k; run --env AntPush-v0 --envs 2048 --opt ls_fast_iterations=$k; run --env AntFall-v0 --envs 2048 --opt ls_fast_iterations=$k; run --env AntMultiPush-v0 --envs 2048 --opt ls_fast_iterations=$k; done
