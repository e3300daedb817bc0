#!/bin/bash
# developer aid: csrc/exp_GENPROF.so — generic_kernels.hip built with the general engine's phase timers (-DMZ_EXP_GENPROF: shader cycles
# per phase of gen_forward / gen_solve, printed by every 97th env), the other objects from the product build.  Selected at run time by
# MZ_LIBMAZESTEP_EXPERIMENT=GENPROF (tools/exp_general_prof.py).
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-math-errno -DMZ_EXPERIMENTS -DMZ_EXP_GENPROF $GENPROF_EXTRA -c -o /tmp/generic_GENPROF.o generic_kernels.hip && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_GENPROF.so mazestep.o ant_kernels.o planar_kernels.o /tmp/generic_GENPROF.o && ls -la exp_GENPROF.so
