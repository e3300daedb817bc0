# after the last kernel change of the round: GPU suite, smoke, bench lines + rocprof / PMC passes of the Ant workloads, phase timers, short soak
cd $GRAFT_REPO_ROOT
E=gpurun_out/evidence_r04; mkdir -p $E
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 2>&1 | tail -6 | tee $E/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $E/smoke.txt
python bench.py > $E/bench_line.json 2> $E/bench_line.err
for w in "AntUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096"; do
  set -- $w
  tools/profile.sh r04 $1 $2 > $E/profile_$1.log 2>&1
  python tools/pmc_summary.py r04 $1 $2 > $E/pmc_summary_$1.log 2>&1
  [ "$1" != "AntUMaze-v0" ] && python bench.py --no-cpu-baseline --env $1 --envs $2 > $E/bench_line_$1_$2.json 2>/dev/null
done
python bench.py > $E/bench_line.json 2> $E/bench_line.err
rm -rf $E/profiles_r04c; mkdir -p $E/profiles_r04c; cp profiles/r04/*Ant* $E/profiles_r04c/
python tools/phase_profile.py 16 2>/dev/null > $E/phase_cycles.txt
python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null > $E/phase_cycles_AntPush-v0_2048.txt
python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $E/tail_phases.txt
python tools/tail_phases.py 32 AntPush-v0 2048 2>/dev/null | grep -v Warning > $E/tail_phases_AntPush-v0_2048.txt
python tools/tail_probe.py 2>/dev/null | grep -v Warning | grep -v "c /=" > $E/load_balance.txt
bash tools/gpu_bench_all.sh
python tools/exp_open_maze.py 2>/dev/null | tail -2 | tee $E/open_maze.txt
for a in "--envs 8192" "--envs 16384" "--envs 32768" "--env AntFall-v0 --envs 2048" "--env AntMultiFall-v0 --envs 2048" "--env AntSmallBilliard-v0 --envs 2048" "--env AntMultiPush-v0 --envs 2048" "--env AntPushMaze-v0 --envs 2048" "--env AntPush-v0 --envs 4096" "--env AntPush-v0 --envs 8192"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done | tee $E/other_configs_ant.txt
timeout 1200 python tools/soak.py 60000 AntUMaze-v0,Ant4Rooms-v0,AntPush-v0,AntFall-v0,AntSmallBilliard-v0 2>/dev/null | tee $E/soak_long_ant.txt
timeout 900 python tools/parity_stats.py 2>/dev/null > $E/parity.md; head -8 $E/parity.md
