cd $GRAFT_REPO_ROOT
export MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/libmazestep_dev.so
mkdir -p gpurun_out/subtick
python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null | tee gpurun_out/subtick/push32.txt
python tools/phase_profile.py 16 AntPush-v0 2048 2>/dev/null | tee gpurun_out/subtick/push16.txt
python tools/phase_profile.py 16 2>/dev/null | tee gpurun_out/subtick/ant16.txt
python tools/tail_phases.py 32 AntPush-v0 2048 2>&1 | grep -v amdgpu | tail -24 | tee gpurun_out/subtick/tail_push.txt
