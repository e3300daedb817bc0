# developer loop for the ant with one movable block: the dev library (make -C mujoco_maze_amd/csrc dev1) through the AntPush / AntFall
# GPU tests, a lanes sweep and (TAIL=1) the per-wave phase breakdown
cd $GRAFT_REPO_ROOT
out=gpurun_out/dev_push; mkdir -p $out
export MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/libmazestep_dev.so
python -m pytest tests/test_gpu_parity.py -q -x -k "ant_push_movable or (ant_fall_maze and AntFall) or MultiFall-v2 or ant_single_step or corner_contacts or (full_size and Push) or record" -p no:cacheprovider 2>&1 | tail -25 > $out/pytest.log; cat $out/pytest.log
for l in 16 32; do python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --env AntPush-v0 --envs 2048 --lanes $l 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(\"AntPush lanes $l: %.3f M env-steps/s kernel %.4f ms bad %d\" % (d[\"value\"]/1e6, d[\"roofline\"][\"kernel_ms\"], d[\"config\"][\"bad_envs\"]))"; done | tee $out/bench.txt
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('AntUMaze: %.3f M env-steps/s kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))" | tee -a $out/bench.txt
if [ -n "$TAIL" ]; then python tools/tail_phases.py ${TAIL} AntPush-v0 2048 2>&1 | grep -v amdgpu | tail -18 | tee $out/tail_phases.txt; fi
