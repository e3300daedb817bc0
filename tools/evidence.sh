#!/bin/bash
# GPU box: the evidence of a round -> gpurun_out/evidence_<tag>/ (what is kept is copied to profiles/<tag>/).
#   part a: the default bench line, rocprofv3 stats + PMC passes per BASELINE workload (tools/profile.sh, tools/pmc_summary.py writes
#           profiles/<tag>/{summary,kernel_stats,pmc}_<env>_<envs>.*), their bench lines, phase timers and tail phases
#   part b: soak, parity quantiles, the configs next to the BASELINE ones, the batch-size sweep
#   tools/evidence.sh <tag> [a|b|ab]
tag=${1:-r06}; part=${2:-ab}
cd $GRAFT_REPO_ROOT
out=gpurun_out/evidence_$tag
mkdir -p $out
if [[ $part == *a* ]]; then
  python bench.py > $out/bench_line.json 2> $out/bench_line.err
  python bench.py --steps 20 --warmup 5 > $out/bench_line_driver_like.json 2>/dev/null
  for w in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096" "AntUMaze-v0 8192"; do
    set -- $w
    tools/profile.sh $tag $1 $2 > $out/profile_$1_$2.log 2>&1
    python tools/pmc_summary.py $tag $1 $2 > $out/pmc_summary_$1_$2.log 2>&1
    rm -rf gpurun_out/prof_${tag}_$1_$2  # the raw rocprofv3 output (tens of MiB per workload): only what pmc_summary.py kept travels back
    [ "$w" != "AntUMaze-v0 4096" ] && python bench.py --no-cpu-baseline --env $1 --envs $2 > $out/bench_line_$1_$2.json 2>/dev/null
  done
  cp -r profiles/$tag $out/profiles_$tag
  python tools/phase_profile.py 16 2>/dev/null > $out/phase_cycles.txt
  python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null > $out/phase_cycles_AntPush-v0_2048.txt
  python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $out/tail_phases.txt
  python tools/tail_phases.py 32 AntPush-v0 2048 2>/dev/null | grep -v Warning > $out/tail_phases_AntPush-v0_2048.txt
  bash tools/fetch_calib.sh > /dev/null 2>&1; cp gpurun_out/calib/fetch_calibration.txt $out/ 2>/dev/null
fi
if [[ $part == *b* ]]; then
  timeout 1800 python tools/soak.py ${SOAK_STEPS:-8000} 2>/dev/null > $out/soak.txt
  timeout 1500 python tools/parity_stats.py ${PARITY_MODE:-} 2>/dev/null > $out/parity.md
  bash tools/gpu_other_configs.sh > /dev/null 2>&1; cp gpurun_out/other_configs.txt $out/
  python tools/general_engine_bench.py 2048 100 2>/dev/null | grep -v Warning > $out/general_engine.txt
  for a in "--envs 8192" "--envs 16384" "--envs 32768" "--env PointUMaze-v0 --envs 8192" "--env PointUMaze-v0 --envs 16384" "--env AntPush-v0 --envs 4096" "--env AntPush-v0 --envs 8192"; do
    python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --sustained 0 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
  done > $out/batch_sweep.txt
  cat $out/soak.txt $out/other_configs.txt $out/batch_sweep.txt
fi
