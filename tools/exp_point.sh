#!/bin/bash
# GPU box helper: A/B timing of experiment builds of the library (csrc/exp_<name>.so, built with -DMZ_EXP_<name>: parts of the
# Point step switched off — wrong physics, timing only) against the product library
cd $GRAFT_REPO_ROOT
echo "product:"; python tools/bench_point.py 0 ${1:-PointUMaze-v0} 2>/dev/null | tail -1
for f in mujoco_maze_amd/csrc/exp_*.so; do
  echo "$f:"; MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/$f python tools/bench_point.py 0 ${1:-PointUMaze-v0} 2>/dev/null | tail -1
done
