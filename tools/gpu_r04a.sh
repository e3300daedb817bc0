# round 4, call A: the round's boundary / fixture additions on the device + baseline numbers (bench with the live PMC passes)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04a; mkdir -p $out
python -m pytest tests -m gpu -q -x > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
tail -15 $out/pytest.log
(time python bench.py --steps 20 --warmup 5) > $out/bench_driver_like.json 2> $out/bench_driver_like.err; tail -3 $out/bench_driver_like.err
python bench.py > $out/bench_line.json 2> $out/bench_line.err
python -c "
import json
for f in ['bench_driver_like','bench_line']:
    d=json.loads(open('$out/'+f+'.json').read().strip().splitlines()[-1]); r=d['roofline']
    print(f, '%.3f M'%(d['value']/1e6), 'kernel_ms %.4f'%r['kernel_ms'], 'traffic', r['traffic'], 'x', r.get('traffic_over_algorithmic'), '|', r['traffic_source'][:90])
    print('   valu', {k:(round(v,4) if isinstance(v,float) else v) for k,v in d.get('roofline_valu',{}).items() if k in ('valu_busy_frac','frac','achieved','peak','flops_per_env_step')})
"
for a in "--env PointUMaze-v0" "--env AntPush-v0 --envs 2048" "--env Ant4Rooms-v0" "--env SwimmerUMaze-v0"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%-62s %8.3f M env-steps/s   kernel %.4f ms   traffic x%.2f (%s)   flagged envs %d' % (d['metric'][34:], d['value']/1e6, r['kernel_ms'], r.get('traffic_over_algorithmic') or 0, r['traffic_source'][:4], d['config']['bad_envs']))"
done > $out/other_configs.txt; cat $out/other_configs.txt
