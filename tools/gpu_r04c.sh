# full GPU suite + bench lines + 8192-env residency check of the plain ant
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --timeout 300 2>&1 | tail -15 | tee gpurun_out/pytest_gpu_r04c.txt
bash tools/gpu_bench_all.sh
for n in 8192 16384; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --env AntUMaze-v0 --envs $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('AntUMaze envs=$n  %8.3f M env-steps/s   kernel %.4f ms  flagged %d' % (d['value']/1e6, r['kernel_ms'], d['config']['bad_envs']))"
done | tee gpurun_out/ant_large.txt
