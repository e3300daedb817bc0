#!/bin/bash
# developer aid: experiment library csrc/exp_<NAME>.so — planar_kernels.hip built with -DMZ_EXP_<NAME> (phase timers: PROF for the bare
# Point, SWPROF for the chain kernels; NOARROW / NOCOLLISION / NONEWTON / NODETECT switch parts of the Point step off: wrong physics,
# timing only), the other objects from the product build.  Selected at run time by MZ_LIBMAZESTEP_EXPERIMENT (tools/exp_*.py|sh).
#   tools/exp_build.sh SWPROF
name=${1:?name}
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" && make libmazestep.so > /dev/null && \
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-math-errno -freciprocal-math -DMZ_EXPERIMENTS -DMZ_EXP_$name -c -o /tmp/planar_$name.o planar_kernels.hip && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o exp_$name.so mazestep.o ant_kernels.o /tmp/planar_$name.o generic_kernels.o && ls -la exp_$name.so
