#!/bin/bash
# GPU box: rocprofv3 stats + PMC passes + tail phases of the multi-block / MultiFall kernels -> profiles/r06
cd $GRAFT_REPO_ROOT
tag=r06
out=gpurun_out/multi_$tag
mkdir -p $out profiles/$tag
for w in "AntPushMaze-v0 2048" "AntMultiPush-v0 2048" "AntMultiFall-v0 2048"; do
  set -- $w
  BENCH_ARGS="--steps 300 --warmup 20" tools/profile.sh $tag $1 $2 > $out/profile_$1_$2.log 2>&1
  python tools/pmc_summary.py $tag $1 $2 > $out/pmc_summary_$1_$2.log 2>&1
  rm -rf gpurun_out/prof_${tag}_$1_$2
  python bench.py --no-cpu-baseline --steps 300 --warmup 20 --env $1 --envs $2 > $out/bench_line_$1_$2.json 2>/dev/null
  python tools/tail_phases.py 64 $1 $2 2>/dev/null | grep -v Warning > $out/tail_phases_$1_$2.txt
done
cp -r profiles/$tag $out/profiles_$tag
ls -la $out $out/profiles_$tag
