# Point developer loop: the Point GPU tests + bench (+ live PMC traffic) + phase timers of the experiment library when it was built
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/point
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x --timeout 120 -k "point or Point or golden or detect" -p no:cacheprovider 2>&1 | tail -8 | tee gpurun_out/point/pytest.txt
for a in "--env PointUMaze-v0" "--env Point4Rooms-v0" "--env PointUMaze-v0 --envs 8192" "--env PointUMaze-v0 --envs 16384"; do
python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-50s %8.3f M env-steps/s   kernel %.4f ms   traffic x%.2f (%s)   flagged envs %d' % (d['metric'][34:], d['value']/1e6, r['kernel_ms'], r.get('traffic_over_algorithmic') or 0, r['traffic_source'][:4], d['config']['bad_envs']))"
done | tee gpurun_out/point/bench.txt
if [ -f mujoco_maze_amd/csrc/exp_PROF.so ]; then timeout 300 python tools/exp_point_prof.py 2>&1 | tee gpurun_out/point/prof.txt; fi
