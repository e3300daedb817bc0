#!/bin/bash
# round-3 GPU call A: test suite, bench (1 GPU, refusal, 2-rank rehearsal), per-workload rocprof evidence
cd $GRAFT_REPO_ROOT
out=gpurun_out/r03a; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --gpus 2 > $out/bench_gpus2_refused.out 2> $out/bench_gpus2_refused.err; echo "rc=$?" >> $out/bench_gpus2_refused.err
MZ_BENCH_SINGLE_GPU=1 python bench.py --gpus 2 --env Ant4Rooms-v0 --steps 300 --warmup 20 > $out/bench_2rank_rehearsal.json 2> $out/bench_2rank_rehearsal.err
for cfg in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096"; do
  tools/profile.sh r03 $cfg > $out/profile_${cfg// /_}.log 2>&1
  python tools/pmc_summary.py r03 $cfg > $out/pmc_summary_${cfg// /_}.log 2>&1
done
mkdir -p $out/profiles_r03; cp -r profiles/r03/* $out/profiles_r03/ 2>/dev/null
tail -3 $out/pytest.log; cat $out/bench_line.json | head -c 600
