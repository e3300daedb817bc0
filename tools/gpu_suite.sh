# the whole GPU suite (with the discontinuity-route log), smoke, and the bench lines of the BASELINE workloads
cd $GRAFT_REPO_ROOT; out=gpurun_out/suite; mkdir -p $out; rm -f $out/parity_log.txt
MZ_PARITY_LOG=$GRAFT_REPO_ROOT/$out/parity_log.txt timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider 2>&1 | tail -15 | tee $out/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $out/smoke.txt
if [ -z "$NOBENCH" ]; then
python bench.py --steps 20 --warmup 5 > $out/bench_driver_like.json 2> $out/bench_driver_like.err; tail -c 1500 $out/bench_driver_like.json
for a in "--env AntUMaze-v0" "--env AntUMaze-v0 --envs 8192" "--env Ant4Rooms-v0" "--env AntPush-v0 --envs 2048" "--env PointUMaze-v0" "--env SwimmerUMaze-v0"; do
python bench.py --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=d.get('sustained') or {}; print('%-55s %8.3f M env-steps/s  kernel %.4f ms  sustained %.3f M (kernel %.4f ms)  flagged %d' % (d['metric'][24:], d['value']/1e6, r['kernel_ms'], s.get('value',0)/1e6, s.get('kernel_ms',0), d['config']['bad_envs']))"
done | tee $out/bench_lines.txt
fi
