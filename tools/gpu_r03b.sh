#!/bin/bash
# round-3 evidence call: everything profiles/r03/ holds, collected on the kernels of the current tree in ONE gpurun call:
# GPU test suite, the default bench line (+ refusal / 2-rank rehearsal of the N > 1 path), rocprofv3 kernel stats + PMC passes per
# BASELINE workload (-> profiles/r03 via tools/pmc_summary.py), bench lines that then carry those counters, in-kernel phase
# timers, per-wave load balance, parity quantiles, throughput of the other configs, soaks.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03b; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
python bench.py --gpus 2 > $out/bench_gpus2_refused.out 2> $out/bench_gpus2_refused.err; echo "rc=$?" >> $out/bench_gpus2_refused.err
MZ_BENCH_SINGLE_GPU=1 python bench.py --gpus 2 --env Ant4Rooms-v0 --steps 300 --warmup 20 > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank_rehearsal.err
for cfg in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096"; do
  tools/profile.sh r03 $cfg > $out/profile_${cfg// /_}.log 2>&1
  python tools/pmc_summary.py r03 $cfg > $out/pmc_summary_${cfg// /_}.log 2>&1
done
# bench lines AFTER the counters of this tree exist (roofline.traffic / roofline_valu read profiles/r03/pmc_<env>_<envs>.csv)
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --env PointUMaze-v0 > $out/bench_line_PointUMaze-v0_4096.json 2>/dev/null
python bench.py --env AntPush-v0 --envs 2048 > $out/bench_line_AntPush-v0_2048.json 2>/dev/null
python bench.py --env Ant4Rooms-v0 > $out/bench_line_Ant4Rooms-v0_4096.json 2>/dev/null
python bench.py --env SwimmerUMaze-v0 > $out/bench_line_SwimmerUMaze-v0_4096.json 2>/dev/null
python tools/phase_profile.py 16 2>/dev/null > $out/phase_cycles.txt
python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null > $out/phase_cycles_AntPush-v0_2048.txt
python tools/tail_probe.py 2>/dev/null | grep -v Warning | grep -v "c /=" > $out/load_balance.txt
for a in "--env Ant4Rooms-v0" "--env AntPush-v0 --envs 2048" "--env PointUMaze-v0" "--env Point4Rooms-v0" "--env AntPushMaze-v0 --envs 2048" "--env AntMultiPush-v0 --envs 2048" \
         "--env AntFall-v0 --envs 2048" "--env AntMultiFall-v0 --envs 2048" "--env AntSmallBilliard-v0 --envs 2048" "--env PointFall-v0" "--env PointPush-v0" "--env PointPushMaze-v0" "--env PointBilliard-v0" \
         "--env SwimmerUMaze-v0" "--env ReacherUMaze-v0" "--envs 8192" "--envs 16384" "--envs 32768"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/other_configs.txt
python tools/bench_generic.py 2>/dev/null > $out/generic_robots.txt
# phase timers of the bare Point (with the per-wave spread) and of the chain kernels: experiment libraries built by tools/exp_build.sh PROF / SWPROF
[ -f mujoco_maze_amd/csrc/exp_PROF.so ] && python tools/exp_point_prof.py PointUMaze-v0 > $out/point_phase_tail.txt 2>&1
[ -f mujoco_maze_amd/csrc/exp_SWPROF.so ] && { python tools/exp_swimmer_prof.py SwimmerUMaze-v0; python tools/exp_swimmer_prof.py ReacherUMaze-v0; } > $out/chain_phase.txt 2>&1
python tools/parity_stats.py 2>/dev/null > $out/parity.md
python tools/soak.py 8000 2>/dev/null > $out/soak.txt
mkdir -p $out/profiles_r03; cp -r profiles/r03/* $out/profiles_r03/ 2>/dev/null
tail -3 $out/pytest.log; head -c 400 $out/bench_line.json; echo; cat $out/other_configs.txt
