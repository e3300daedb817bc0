# A/B: HBM traffic of the Point step with and without the XCD-aware block map (tools/exp_build.sh NOXCD)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lib in "" "$R/mujoco_maze_amd/csrc/exp_NOXCD.so"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/px; MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$lib rocprofv3 --pmc $c --output-format csv -d /tmp/px -- python $R/bench.py --no-cpu-baseline --no-live-pmc --env PointUMaze-v0 --steps 100 --warmup 0 > /tmp/px.log 2>&1
    python - "$c" "${lib:-product}" <<'PY'
import csv, glob, sys
v=[float(r["Counter_Value"]) for f in glob.glob("/tmp/px/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f)) if "planar_step" in r["Kernel_Name"] and r["Counter_Name"]==sys.argv[1]]
print(sys.argv[2][-14:], sys.argv[1], "launches", len(v), "mean KB", sum(v[100:])/max(1,len(v[100:])))
PY
  done
  MZ_DEBUG=1 MZ_LIBMAZESTEP_EXPERIMENT=$lib python $R/bench.py --no-cpu-baseline --no-live-pmc --env PointUMaze-v0 --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   %.3f M env-steps/s kernel %.4f ms' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
