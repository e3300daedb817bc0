# How does the live PMC traffic scale with the batch?  (fixed part = instruction fetch of the kernel's code by each of the 8 XCDs' L2)
cd $GRAFT_REPO_ROOT
envid=${1:-PointUMaze-v0}
for n in 1024 2048 4096 8192 16384; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --env $envid --envs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$envid envs $n %8.3f M  kernel %.4f ms  algorithmic %d  fetch %d B  write %d B  x%.2f' % (d['value']/1e6, r['kernel_ms'], r['algorithmic_bytes_per_launch'], r['traffic_fetch_bytes'], r['traffic_write_bytes'], r['traffic_over_algorithmic']))"
done | tee -a gpurun_out/fetch_sweep.txt
