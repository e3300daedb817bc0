#!/bin/bash
# developer aid: ISA of ONE instantiation of ant_step_kernel (seconds, not minutes) -> /tmp/isa/k_<NB>_<G>.s, resource usage on stdout
#   tools/isa_one.sh [NB [G]]
NB=${1:-0}; G=${2:-16}; PROF=${3:-false}; WPS=${4:-0}  # fourth argument 2: the two-waves-per-SIMD instantiation   # third argument "true": the instrumented build (s_memtime marks the phase boundaries)
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" && mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-math-errno -freciprocal-math -fno-signed-zeros -fapprox-func \
  -DMZ_ISA_ONLY -DMZ_ISA_NB=$NB -DMZ_ISA_G=$G -DMZ_ISA_PROF=$PROF -DMZ_ISA_WPS=$WPS ${EXTRA:+-DMZ_EXPERIMENTS} $EXTRA --cuda-device-only -S -o /tmp/isa/k_${NB}_${G}.s ant_kernels.hip 2>&1 | grep -v "hip-link"
grep -E "^\s*; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)" /tmp/isa/k_${NB}_${G}.s | head -8
