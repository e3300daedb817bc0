# GPU box (experiment): lanes per env of the multi-block / ball ant mazes (lane-group solver)
cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-64s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
for e in AntMultiPush-v0 AntPushMaze-v0 AntSmallBilliard-v0 AntMultiFall-v0; do for n in 2048 4096; do for l in 64 32 16; do run --env $e --envs $n --lanes $l; done; done; done
