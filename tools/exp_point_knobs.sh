cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 500 --warmup 20 --no-cpu-baseline --no-live-pmc --sustained 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-64s %.3f M env-steps/s  kernel %.4f ms  bad %d' % ('$*', d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"; }
for l in 32 16 8; do run --env PointUMaze-v0 --envs 4096 --lanes $l; run --env PointUMaze-v0 --envs 8192 --lanes $l; done
for w in 2 4; do run --env PointUMaze-v0 --envs 4096 --wpb $w; done
run --env Point4Rooms-v0 --envs 4096 --lanes 16
run --env Point4Rooms-v0 --envs 4096 --lanes 32
