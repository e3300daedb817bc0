#!/bin/bash
# GPU box helper (round 4, part B): soak, parity quantiles, the other configs
tag=${1:-r04}
cd $GRAFT_REPO_ROOT
out=gpurun_out/evidence_$tag
mkdir -p $out
timeout 1500 python tools/soak.py ${SOAK_STEPS:-8000} 2>/dev/null > $out/soak.txt
timeout 900 python tools/parity_stats.py 2>/dev/null > $out/parity.md
for a in "--env AntPushMaze-v0 --envs 2048" "--env AntMultiPush-v0 --envs 2048" \
         "--env AntFall-v0 --envs 2048" "--env AntMultiFall-v0 --envs 2048" "--env AntSmallBilliard-v0 --envs 2048" "--env PointFall-v0" "--env PointPush-v0" "--env PointPushMaze-v0" "--env PointBilliard-v0" \
         "--env Point4Rooms-v0" "--env ReacherUMaze-v0" "--envs 8192" "--envs 16384" "--envs 32768" "--env PointUMaze-v0 --envs 8192" "--env PointUMaze-v0 --envs 16384" "--env AntPush-v0 --envs 4096" "--env AntPush-v0 --envs 8192"; do
  python bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-live-pmc $a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-62s %8.3f M env-steps/s   kernel %.4f ms   flagged envs %d' % (d['metric'][34:], d['value']/1e6, d['roofline']['kernel_ms'], d['config']['bad_envs']))"
done > $out/other_configs.txt
cat $out/soak.txt $out/other_configs.txt
