# the bench lines of the BASELINE configs (300 timed steps each, live PMC traffic) -> gpurun_out/bench_all.txt
cd $GRAFT_REPO_ROOT
for a in "--env AntUMaze-v0" "--env PointUMaze-v0" "--env AntPush-v0 --envs 2048" "--env Ant4Rooms-v0" "--env SwimmerUMaze-v0"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-62s %8.3f M env-steps/s   kernel %.4f ms   traffic x%.2f (%s)   flagged envs %d' % (d['metric'][34:], d['value']/1e6, r['kernel_ms'], r.get('traffic_over_algorithmic') or 0, r['traffic_source'][:4], d['config']['bad_envs']))"
done | tee gpurun_out/bench_all.txt
