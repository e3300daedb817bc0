#!/bin/bash
# GPU box: the "rk4 slot of the slowest waves" (VERDICT r05 #3b) with the instrumented kernel's same-address atomics removed: phase
# timers with and without the rk4 sub-timers (dev libraries: make dev [DEVFLAGS=-DMZ_EXP_RK4TICK])
cd $GRAFT_REPO_ROOT; out=gpurun_out/rk4_r06; mkdir -p $out
export MZ_DEBUG=1
MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/libmazestep_dev.so python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $out/tail_phases_noatomics.txt
MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/exp_ant_rk4tick.so python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $out/tail_phases_noatomics_rk4tick.txt
MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/libmazestep_dev.so python tools/phase_profile.py 16 2>/dev/null | grep -v Warning > $out/phase_cycles.txt
head -20 $out/tail_phases_noatomics.txt; head -22 $out/tail_phases_noatomics_rk4tick.txt; cat $out/phase_cycles.txt
