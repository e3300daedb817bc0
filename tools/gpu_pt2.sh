cd $GRAFT_REPO_ROOT
export MZ_LIBMAZESTEP_EXPERIMENT=$GRAFT_REPO_ROOT/mujoco_maze_amd/csrc/exp_PT2.so
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -x --timeout 120 -k "point or Point or golden or detect" -p no:cacheprovider 2>&1 | tail -4
for a in "--env PointUMaze-v0" "--env Point4Rooms-v0" "--env PointPush-v0" "--env PointUMaze-v0 --envs 8192"; do
python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-50s %8.3f M env-steps/s   kernel %.4f ms  flagged envs %d' % (d['metric'][34:], d['value']/1e6, r['kernel_ms'], d['config']['bad_envs']))"
done
python tools/exp_point_prof.py 2>/dev/null | head -3
