cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ant_push_movable or full_size" -p no:cacheprovider --timeout 300 2>&1 | tail -4
for a in "--envs 2048" "--envs 4096" "--envs 8192" "--envs 4096 --lanes 32"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --env AntPush-v0 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AntPush $a  %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done | tee gpurun_out/push_lanes.txt
for a in "--env AntFall-v0 --envs 4096"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$a  %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done | tee -a gpurun_out/push_lanes.txt
