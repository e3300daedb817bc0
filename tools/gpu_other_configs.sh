# bench lines of the configs next to the BASELINE ones (block / ball / fall mazes; 300 timed steps each) -> gpurun_out/other_configs.txt
cd $GRAFT_REPO_ROOT
for a in "AntMultiPush-v0 2048" "AntPushMaze-v0 2048" "AntFall-v0 2048" "AntMultiFall-v0 2048" "AntSmallBilliard-v0 2048" "Point4Rooms-v0 4096" "PointPush-v0 4096" "PointPushMaze-v0 4096" "PointBilliard-v0 4096" "PointFall-v0 4096" "ReacherUMaze-v0 4096"; do
  set -- $a
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --sustained 0 --env $1 --envs $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-30s %6d envs %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % ('$1', $2, d['value']/1e6, r['kernel_ms'], d['config']['bad_envs']))"
done | tee gpurun_out/other_configs.txt
