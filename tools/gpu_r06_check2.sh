cd $GRAFT_REPO_ROOT; out=gpurun_out/check2_r06; mkdir -p $out
python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > $out/gpu_suite.log; cat $out/gpu_suite.log
for w in "AntUMaze-v0 4096" "PointUMaze-v0 4096" "AntPush-v0 2048" "AntUMaze-v0 8192"; do set -- $w
  python bench.py --no-cpu-baseline --no-live-pmc --steps 500 --warmup 20 --sustained 0 --env $1 --envs $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 guard off %.3f M kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
  python bench.py --no-cpu-baseline --no-live-pmc --steps 500 --warmup 20 --sustained 0 --env $1 --envs $2 --opt ls_guard=1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1 $2 guard ON  %.3f M kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done | tee $out/guard_onoff.txt
MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/exp_ant_rk4tick.so python tools/tail_phases.py 16 2>/dev/null | grep -v Warning > $out/tail_phases_rk4tick_b.txt; cat $out/tail_phases_rk4tick_b.txt | head -22
python tools/tail_phases.py 16 AntPush-v0 2048 2>/dev/null | grep -v Warning > $out/tail_phases_AntPush_16.txt; head -18 $out/tail_phases_AntPush_16.txt
python tools/tail_phases.py 32 AntPush-v0 2048 2>/dev/null | grep -v Warning > $out/tail_phases_AntPush_32.txt; head -18 $out/tail_phases_AntPush_32.txt
