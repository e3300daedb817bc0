cd $GRAFT_REPO_ROOT
out=gpurun_out/r03c; mkdir -p $out
python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -3 $out/pytest.log
python tools/exp_point_prof.py PointUMaze-v0 > $out/point_phase.txt 2>&1; cat $out/point_phase.txt
for cfg in "SwimmerUMaze-v0 4096"; do
  tools/profile.sh r03 $cfg > $out/profile_${cfg// /_}.log 2>&1
  python tools/pmc_summary.py r03 $cfg > $out/pmc_summary_${cfg// /_}.log 2>&1
done
python bench.py --env SwimmerUMaze-v0 > $out/bench_line_SwimmerUMaze-v0_4096.json 2>/dev/null
for a in "--env SwimmerUMaze-v0" "--env ReacherUMaze-v0" "--env Swimmer4Rooms-v0" "--env PointUMaze-v0"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/chain_configs.txt; cat $out/chain_configs.txt
python tools/parity_stats.py 2>/dev/null > $out/parity.md; grep -i "swimmer\|reacher" $out/parity.md
MZ_SOAK_ONLY=chain python tools/soak.py 8000 2>/dev/null > $out/soak.txt; grep -i "swimmer\|reacher" $out/soak.txt
mkdir -p $out/profiles_r03; cp profiles/r03/*Swimmer* $out/profiles_r03/
