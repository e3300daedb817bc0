# developer loop for the plain ant's kernel: the dev library (make -C mujoco_maze_amd/csrc dev) through the plain-ant GPU tests,
# a short bench and (PROF=1) the phase timers
cd $GRAFT_REPO_ROOT
out=gpurun_out/dev_ant; mkdir -p $out
export MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/libmazestep_dev.so
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 200 -k "ant_single_step_parity or ant_forward_dynamics or lane_group_widths or wall_contacts_and_goal or corner_contacts or lanes_of_a_row or (full_size and not Push) or subgoal or rollout_tracks or ragged" -p no:cacheprovider 2>&1 | tail -25 > $out/pytest.log; cat $out/pytest.log
python bench.py --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AntUMaze 1000 steps: %.3f M env-steps/s  kernel %.4f ms  bad %d' % (d['value']/1e6, r['kernel_ms'], d['config']['bad_envs']))" | tee $out/bench.txt
python bench.py --no-cpu-baseline --no-live-pmc --env Ant4Rooms-v0 --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('Ant4Rooms 300 steps: %.3f M env-steps/s  kernel %.4f ms  bad %d' % (d['value']/1e6, r['kernel_ms'], d['config']['bad_envs']))" | tee -a $out/bench.txt
if [ -n "$PROF" ]; then python tools/phase_profile.py 16 AntUMaze-v0 4096 2>&1 | tail -20 | tee $out/phase.txt; fi
if [ -n "$TAIL" ]; then python tools/tail_phases.py 16 AntUMaze-v0 4096 2>&1 | tail -20 | tee $out/tail_phases.txt; fi
