#!/usr/bin/env python3
"""Summarise rocprofv3 outputs under gpurun_out/prof_<tag> into profiles/<tag>_summary.md (+ trimmed CSV)."""
import csv, glob, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs(f"profiles/{tag}", exist_ok=True)
def newest(pattern):
    """gpurun merges a call's outputs next to those of earlier calls: keep only the latest file of each pass directory."""
    by_dir = {}
    for f in glob.glob(pattern, recursive=True):
        top = os.path.relpath(f, src).split(os.sep)[0]
        if top not in by_dir or os.path.getmtime(f) > os.path.getmtime(by_dir[top]): by_dir[top] = f
    return sorted(by_dir.values())
lines = [f"# rocprofv3 summary — {tag}", "", "command: `python bench.py --no-cpu-baseline` (= the default bench run: 100 settle + 100 warm-up + 1000 timed steps) (AntUMaze-v0, 4096 envs, 1 x MI355X)", ""]
for f in newest(f"{src}/stats/**/*kernel_stats.csv"):
    lines += ["## kernel stats (`rocprofv3 --kernel-trace --stats`)", "", "| kernel | calls | avg us | min us | max us | % |", "|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        name = r["Name"].split("(")[0][:60]
        lines.append(f"| {name} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {float(r['Percentage']):.2f} |")
    lines.append("")
# the bench line times the launches after the 100 settle + 100 warm-up steps: same window from the kernel trace
WARM = 200
for f in newest(f"{src}/stats/**/*kernel_trace.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "ant_step_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    if len(dur) > WARM:
        timed = dur[WARM:]
        lines += [f"`ant_step_kernel` over the timed window (launches {WARM + 1}..{len(dur)}): avg **{sum(timed)/len(timed):.1f} us** "
                  f"(min {min(timed):.1f}, max {max(timed):.1f}); the {WARM} untimed settle / warm-up launches avg {sum(dur[:WARM])/WARM:.1f} us "
                  "(early in the rollout the ants are still airborne / settling: fewer contacts).", ""]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in newest(f"{src}/pmc_*/**/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for extra in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
            if extra in r: agg[k]["_" + extra] = [float(r[extra])]
lines += ["## PMC counters (separate `--pmc` passes), per-launch averages", ""]
for k, d in agg.items():
    if "ant_step" not in k and "point_step" not in k: continue
    lines += [f"### {k[:70]}", "", "| counter | avg per launch | launches |", "|---|---|---|"]
    for c, v in sorted(d.items()):
        lines.append(f"| {c} | {sum(v)/len(v):.4g} | {len(v)} |")
    lines.append("")
open(f"profiles/{tag}/summary.md", "w").write("\n".join(lines))
print("\n".join(lines))
# raw material next to the summary
import shutil
for pat, dst in (("stats/**/*kernel_stats.csv", "kernel_stats.csv"), ("stats/**/*domain_stats.csv", "domain_stats.csv")):
    for f in newest(f"{src}/{pat}"):
        shutil.copy(f, f"profiles/{tag}/{dst}")
with open(f"profiles/{tag}/pmc_ant_step_kernel.csv", "w") as f:
    f.write("counter,avg_per_launch,launches\n")
    for k, d in agg.items():
        if "ant_step_kernel" in k:
            for c, v in sorted(d.items()):
                if not c.startswith("_"):
                    f.write(f"{c},{sum(v)/len(v):.6g},{len(v)}\n")
