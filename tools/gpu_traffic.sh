# fetch / write split of the live PMC traffic per BASELINE workload
cd $GRAFT_REPO_ROOT
for a in "--env PointUMaze-v0" "--env AntPush-v0 --envs 2048" "--env AntUMaze-v0"; do
python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-40s %8.3f M  kernel %.4f ms  algorithmic %d B  fetch %d B  write %d B  x%.2f' % (d['metric'][34:], d['value']/1e6, r['kernel_ms'], r['algorithmic_bytes_per_launch'], r['traffic_fetch_bytes'], r['traffic_write_bytes'], r['traffic_over_algorithmic']))"
done | tee gpurun_out/traffic_split.txt
