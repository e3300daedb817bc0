# after the scratch fix: GPU suite, AntPush / PointPush lines, AntPush PMC passes, long soak + long parity
cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence_r04; mkdir -p $E
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -6 | tee $E/pytest_gpu.txt
bash tools/gpu_bench_all.sh
tools/profile.sh r04 AntPush-v0 2048 > $E/profile_AntPush-v0.log 2>&1
python tools/pmc_summary.py r04 AntPush-v0 2048 > $E/pmc_summary_AntPush-v0.log 2>&1
python bench.py --no-cpu-baseline --env AntPush-v0 --envs 2048 > $E/bench_line_AntPush-v0_2048.json 2>/dev/null
mkdir -p $E/profiles_r04b; cp profiles/r04/*AntPush* $E/profiles_r04b/
for a in "--env PointPush-v0" "--env PointPushMaze-v0" "--env AntFall-v0 --envs 2048" "--env AntMultiFall-v0 --envs 2048"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done | tee $E/other_configs_b.txt
timeout 2400 python tools/soak.py 60000 2>/dev/null | tee $E/soak_long.txt
timeout 1500 python tools/parity_stats.py long 2>/dev/null > $E/parity_long.md; tail -20 $E/parity_long.md
