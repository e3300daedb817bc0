#!/bin/bash
# GPU box: the whole GPU suite, the bench lines of the five BASELINE workloads and the seam experiment of parity_stats (round 6 mid-round check)
cd $GRAFT_REPO_ROOT
out=gpurun_out/check_r06
mkdir -p $out
python -m pytest tests -q -m gpu 2>&1 | tail -12 > $out/gpu_suite.log
python bench.py --no-cpu-baseline --no-live-pmc > $out/bench_line.json 2> $out/bench_line.err
for w in "PointUMaze-v0 4096" "AntPush-v0 2048" "Ant4Rooms-v0 4096" "SwimmerUMaze-v0 4096" "AntUMaze-v0 8192"; do
  set -- $w
  python bench.py --no-cpu-baseline --no-live-pmc --steps 300 --warmup 20 --sustained 0 --env $1 --envs $2 > $out/bench_line_$1_$2.json 2>/dev/null
done
MZ_PS_CONFIGS="AntMultiFall-v0@8,AntFall-v0@2,AntMultiFall-v0,AntFall-v0" python tools/parity_stats.py long 2>/dev/null > $out/parity_seams.md
cat $out/gpu_suite.log; cat $out/parity_seams.md
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/check_r06/bench_line*.json")):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value']/1e6,3), 'M', d['roofline']['kernel_ms'], 'ms', (d.get('sustained') or {}).get('value'), d['config'].get('launch'))
    except Exception as e: print(f, 'ERR', e)
PY
