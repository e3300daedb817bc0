#!/bin/bash
# GPU box, end of round 6: the GPU suite on the final library, then evidence part b (soak, parity, other configs, general engine, batch sweep)
# and the Point's profile on its new record
cd $GRAFT_REPO_ROOT; out=gpurun_out/final_r06; mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -6 > $out/gpu_suite.log; cat $out/gpu_suite.log
tools/profile.sh r06 PointUMaze-v0 4096 > $out/profile_Point.log 2>&1; python tools/pmc_summary.py r06 PointUMaze-v0 4096 > $out/pmc_summary_Point.log 2>&1; rm -rf gpurun_out/prof_r06_PointUMaze-v0_4096
python bench.py --no-cpu-baseline --env PointUMaze-v0 --envs 4096 > $out/bench_line_PointUMaze-v0_4096.json 2>/dev/null
cp profiles/r06/summary_PointUMaze-v0_4096.md profiles/r06/pmc_PointUMaze-v0_4096.csv profiles/r06/kernel_stats_PointUMaze-v0_4096.csv $out/
PARITY_MODE=long bash tools/evidence.sh r06 b
