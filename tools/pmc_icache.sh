#!/bin/bash
# GPU box helper: instruction-cache counters of one bench workload (own --pmc passes, no tracing domains) -> gpurun_out/icache_<env>_<envs>.txt
#   tools/pmc_icache.sh [<env id> [<envs>]]
envid=${1:-AntUMaze-v0}; nenv=${2:-4096}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_icache; res=$GRAFT_REPO_ROOT/gpurun_out/icache_${envid}_${nenv}.txt
mkdir -p $out; cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" | sort -u > $res
CMD="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-live-pmc --sustained 0 --steps 300 --env $envid --envs $nenv"
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  d=$out/$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $d -- $CMD > $d.log 2>&1
  python - "$d" >> $res <<'PY'
import sys, glob, csv, collections
tot = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(tot): print(f"{k}: {tot[k] / max(1, cnt[k]):.4g} per launch ({cnt[k]} rows)")
PY
done
rm -rf $out
cat $res
