#!/bin/bash
# GPU box helper: GPU test suite + short bench lines of the BASELINE configs + in-kernel phase timers -> gpurun_out/<tag>
tag=${1:-perf}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
python bench.py --no-cpu-baseline > $out/bench_line.json 2> $out/bench_line.err
for a in "--env Ant4Rooms-v0" "--env AntPush-v0 --envs 2048" "--env PointUMaze-v0" "--env SwimmerUMaze-v0" ${EXTRA_CFG}; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/configs.txt 2>&1
python tools/phase_profile.py 16 2>/dev/null > $out/phase_cycles.txt
python tools/phase_profile.py 32 AntPush-v0 2048 2>/dev/null > $out/phase_cycles_antpush.txt
tail -5 $out/pytest.log; python -c "import json; d=json.load(open('$out/bench_line.json')); print('AntUMaze default: %.3f M env-steps/s, kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"; cat $out/configs.txt $out/phase_cycles.txt $out/phase_cycles_antpush.txt
