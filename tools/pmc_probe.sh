#!/bin/bash
# GPU box helper: extra PMC passes of a short bench run; prints per-launch averages of the step kernel.
#   tools/pmc_probe.sh "CTR_A CTR_B ..." ["CTR_C ..."]   (one rocprofv3 --pmc pass per argument)
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_probe
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
k=0
for grp in "$@"; do
  k=$((k+1))
  rocprofv3 --pmc $grp --output-format csv -d $out/p$k -- python $GRAFT_REPO_ROOT/bench.py --steps 300 --warmup 10 --no-cpu-baseline ${BENCH_ARGS:-} > $out/p$k.log 2>&1
done
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "step_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:32s} {sum(v)/len(v):14.6g}  ({len(v)} launches)")
PY
