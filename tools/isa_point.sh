#!/bin/bash
# developer aid: ISA of the bare Point's step kernel only (planar_step_kernel<0,0,32>) -> /tmp/isa/point.s, resource usage on stdout
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" && mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-unused-function -fno-math-errno -freciprocal-math -DMZ_ISA_POINT ${EXTRA:+-DMZ_EXPERIMENTS} $EXTRA --cuda-device-only -S -o /tmp/isa/point.s planar_kernels.hip 2>&1 | grep -v "hip-link"
grep -E "^\s*; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize|codeLenInByte)|vgpr_spill_count|sgpr_spill_count" /tmp/isa/point.s | tail -8
echo "scratch instructions: $(grep -c scratch_ /tmp/isa/point.s)"
