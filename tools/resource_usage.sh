#!/bin/bash
# Register / scratch / occupancy report of the step kernels (cross-compile, no GPU needed): tools/resource_usage.sh [grep pattern]
cd "$(dirname "$0")/../mujoco_maze_amd/csrc" || exit 1
FLAGS=$(grep '^HIPFLAGS' Makefile | sed 's/.*?= //; s/\$(ARCH)/gfx950/; s/-Wall//')
/opt/rocm/bin/hipcc $FLAGS -Rpass-analysis=kernel-resource-usage -o /tmp/mz_ru.so mazestep.hip 2>&1 \
  | grep -E "error|Function Name|VGPRs:|AGPRs|ScratchSize|Occupancy" | paste - - - - - \
  | sed 's/remark: [^ ]* //g; s/\[-Rpass-analysis=kernel-resource-usage\]//g; s/mazestep.hip:[0-9]*:[0-9]*://g' | grep -E "${1:-step_kernel}"
