#!/bin/bash
# GPU box: build and run tools/mfma_fold_bench.hip -> gpurun_out/mfma_fold_bench.txt
cd $GRAFT_REPO_ROOT && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -freciprocal-math -fno-signed-zeros -Iinclude -o /tmp/mfma_fold_bench tools/mfma_fold_bench.hip && /tmp/mfma_fold_bench | tee gpurun_out/mfma_fold_bench.txt
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-result -Wno-unused-value -freciprocal-math -fno-signed-zeros -Iinclude -DPAD_PRE -o /tmp/mfma_fold_bench_pad tools/mfma_fold_bench.hip && (echo 'with 4 wait states between fold_t (inline asm) and the MFMAs that read its results:'; /tmp/mfma_fold_bench_pad) | tee -a gpurun_out/mfma_fold_bench.txt
