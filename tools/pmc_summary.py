#!/usr/bin/env python3
"""Summarise the rocprofv3 outputs of tools/profile.sh (gpurun_out/prof_<tag>_<env>_<envs>) into profiles/<tag>/:
summary_<env>_<envs>.md, kernel_stats_<env>_<envs>.csv and pmc_<env>_<envs>.csv (what bench.py's roofline reads).

    python tools/pmc_summary.py <tag> [<env id> [<envs>]]
"""
import collections
import csv
import glob
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
envid = sys.argv[2] if len(sys.argv) > 2 else "AntUMaze-v0"
nenv = sys.argv[3] if len(sys.argv) > 3 else "4096"
key = f"{envid}_{nenv}"
src = f"gpurun_out/prof_{tag}_{key}"
os.makedirs(f"profiles/{tag}", exist_ok=True)


def newest(pattern):
    """gpurun merges a call's outputs next to those of earlier calls: keep only the latest file of each pass directory."""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        top = os.path.relpath(f, src).split(os.sep)[0]
        if top not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[top]):
            by_dir[top] = f
    return sorted(by_dir.values())


lines = [f"# rocprofv3 summary — {tag}, {envid}, {nenv} envs", "",
         f"command: `python bench.py --no-cpu-baseline --env {envid} --envs {nenv}` (100 settle + 100 warm-up + 1000 timed steps, 1 x MI355X)", ""]
for f in newest(f"{src}/stats/**/*kernel_stats.csv"):
    lines += ["## kernel stats (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0][:60]
        lines.append(f"| {name} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines.append("")
# the bench line times the launches after the 100 settle + 100 warm-up steps: same window from the kernel trace
WARM = 200
for f in newest(f"{src}/stats/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if len(dur) > WARM:
        timed = dur[WARM:]
        lines += [f"step kernel over the timed window (launches {WARM + 1}..{len(dur)}): avg **{sum(timed)/len(timed):.1f} us** "
                  f"(min {min(timed):.1f}, max {max(timed):.1f}); the {WARM} untimed settle / warm-up launches avg {sum(dur[:WARM])/WARM:.1f} us.", ""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest(f"{src}/pmc_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if extra in r:
                agg[k]["_" + extra] = [float(r[extra])]
lines += ["## PMC counters (separate `--pmc` passes), per-launch averages", ""]
for k, d in agg.items():
    if "_step_kernel" not in k:
        continue
    lines += [f"### {k[:90]}", "", "| counter | avg per launch | launches |", "|---|---|---|"]
    for c, v in sorted(d.items()):
        lines.append(f"| {c} | {sum(v)/len(v):.4g} | {len(v)} |")
    lines.append("")
open(f"profiles/{tag}/summary_{key}.md", "w").write("\n".join(lines))
print("\n".join(lines))
for pat, dst in (("stats/**/*kernel_stats.csv", f"kernel_stats_{key}.csv"),):
    for f in newest(f"{src}/{pat}"):
        shutil.copy(f, f"profiles/{tag}/{dst}")
with open(f"profiles/{tag}/pmc_{key}.csv", "w") as f:
    f.write("counter,avg_per_launch,launches\n")
    for k, d in agg.items():
        if "_step_kernel" in k:
            for c, v in sorted(d.items()):
                if not c.startswith("_"):
                    f.write(f"{c},{sum(v)/len(v):.6g},{len(v)}\n")
