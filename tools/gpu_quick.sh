#!/bin/bash
# GPU box helper: test suite + short bench lines of the BASELINE configs + resource usage -> gpurun_out/<tag>
tag=${1:-quick}
cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
python -m pytest tests -m gpu -q ${PYTEST_ARGS:-} > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
for a in "" "--env Ant4Rooms-v0" "--env AntPush-v0 --envs 2048" "--env PointUMaze-v0" "--env PointPush-v0" "--env AntFall-v0 --envs 2048" "--env SwimmerUMaze-v0"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/configs.txt 2>&1
tail -25 $out/pytest.log; cat $out/configs.txt
