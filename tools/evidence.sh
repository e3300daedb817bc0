#!/bin/bash
# GPU box helper: everything profiles/<tag>/ holds, in one call: rocprofv3 stats + PMC passes of the default bench command,
# the bench line itself, in-kernel phase timers, per-wave load balance, parity quantiles, the other configs, the soak.
tag=${1:-r02}
cd $GRAFT_REPO_ROOT
out=gpurun_out/evidence_$tag
mkdir -p $out
python bench.py > $out/bench_line.json 2> $out/bench_line.err
tools/profile.sh $tag > $out/profile.log 2>&1
python tools/pmc_summary.py $tag > $out/pmc_summary.log 2>&1
python tools/phase_profile.py 16 2>/dev/null > $out/phase_cycles.txt
python tools/tail_probe.py 2>/dev/null | grep -v Warning | grep -v "c /=" > $out/load_balance.txt
python tools/parity_stats.py 2>/dev/null > $out/parity.md
for a in "--env Ant4Rooms-v0" "--env AntPush-v0 --envs 2048" "--env PointUMaze-v0" "--env AntPushMaze-v0 --envs 2048" "--env AntMultiPush-v0 --envs 2048" \
         "--env AntFall-v0 --envs 2048" "--env AntMultiFall-v0 --envs 2048" "--env AntSmallBilliard-v0 --envs 2048" "--env PointFall-v0" "--env PointPush-v0" "--env PointPushMaze-v0" "--env PointBilliard-v0" \
         "--env SwimmerUMaze-v0" "--env ReacherUMaze-v0" "--envs 8192" "--envs 16384" "--envs 32768"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/other_configs.txt
python tools/soak.py 8000 2>/dev/null > $out/soak.txt
ls -la $out
