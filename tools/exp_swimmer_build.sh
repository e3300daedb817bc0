#!/bin/bash
# developer aid: experiment library of the chain kernels' phase timers (csrc/exp_SWPROF.so: planar_kernels.hip built -DMZ_EXP_SWPROF, the other
# objects from the product build) for tools/exp_swimmer_prof.py
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" && make libmazestep.so > /dev/null && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-math-errno -freciprocal-math -DMZ_EXP_SWPROF -c -o /tmp/planar_swprof.o planar_kernels.hip && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_SWPROF.so mazestep.o ant_kernels.o /tmp/planar_swprof.o generic_kernels.o && ls -la exp_SWPROF.so
