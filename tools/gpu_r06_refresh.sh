#!/bin/bash
# GPU box: bench lines of every workload after the bench's event sampling (time_kernels_stride), GPU suite, other configs, batch sweep
cd $GRAFT_REPO_ROOT; out=gpurun_out/refresh_r06; mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/gpu_suite.log; cat $out/gpu_suite.log
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python bench.py --steps 20 --warmup 5 > $out/bench_line_driver_like.json 2>/dev/null
for w in "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096" "AntUMaze-v0 8192" "AntMultiPush-v0 2048" "AntPushMaze-v0 2048" "AntMultiFall-v0 2048"; do
  set -- $w
  python bench.py --no-cpu-baseline --env $1 --envs $2 > $out/bench_line_$1_$2.json 2>/dev/null
done
bash tools/gpu_other_configs.sh > /dev/null 2>&1; cp gpurun_out/other_configs.txt $out/
for a in "--envs 8192" "--envs 16384" "--envs 32768" "--env PointUMaze-v0 --envs 8192" "--env PointUMaze-v0 --envs 16384" "--env AntPush-v0 --envs 4096" "--env AntPush-v0 --envs 8192"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc --sustained 0 $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/batch_sweep.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/refresh_r06/bench_line*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print('%-44s %8.3f M  ms/step %.4f  kernel %.4f  sustained %.3f M' % (f.split('/')[-1], d['value']/1e6, d['ms_per_step'], r['kernel_ms'], (d.get('sustained') or {}).get('value',0)/1e6))
PY
cat $out/other_configs.txt $out/batch_sweep.txt
