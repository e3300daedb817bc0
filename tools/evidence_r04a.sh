#!/bin/bash
# GPU box helper (round 4, part A): bench lines, rocprofv3 stats + PMC passes per BASELINE workload, phase timers, tail phases
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
out=gpurun_out/evidence_$tag
mkdir -p $out
python bench.py > $out/bench_line.json 2> $out/bench_line.err
for w in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096"; do
  set -- $w
  tools/profile.sh $tag $1 $2 > $out/profile_$1.log 2>&1
  python tools/pmc_summary.py $tag $1 $2 > $out/pmc_summary_$1.log 2>&1
  [ "$1" != "AntUMaze-v0" ] && python bench.py --no-cpu-baseline --env $1 --envs $2 > $out/bench_line_$1_$2.json 2>/dev/null
done
cp -r profiles/$tag $out/profiles_$tag
python tools/phase_profile.py 16 2>/dev/null > $out/phase_cycles.txt
python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null > $out/phase_cycles_AntPush-v0_2048.txt
python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $out/tail_phases.txt
python tools/tail_phases.py 32 AntPush-v0 2048 2>/dev/null | grep -v Warning > $out/tail_phases_AntPush-v0_2048.txt
python tools/tail_probe.py 2>/dev/null | grep -v Warning | grep -v "c /=" > $out/load_balance.txt
ls -la $out
