#!/usr/bin/env python3
"""GPU box helper: phase timers of the bare Point kernel (experiment library csrc/exp_PROF.so, built -DMZ_EXP_PROF): mean shader
cycles per env.step and phase over the printed workgroups of the last 200 steps."""
import os, re, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, MZ_DEBUG="1", MZ_LIBMAZESTEP_EXPERIMENT=os.path.join(root, "mujoco_maze_amd/csrc/exp_PROF.so"))
code = """
import torch, mujoco_maze_amd as mm
n=4096
env = mm.make('%s', num_envs=n, auto_reset=True, force_vec=True); env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, 2), device=env.device, generator=g) * 2 - 1) for _ in range(16)]
for a in acts: a[:, 1] *= 0.25
for i in range(300): env.step(acts[i %% 16])
torch.cuda.synchronize()
""" % (sys.argv[1] if len(sys.argv) > 1 else "PointUMaze-v0")
out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True).stdout
rows = [l for l in out.splitlines() if l.startswith("PROF")]
rows = rows[len(rows) // 3:]
tot = collections.OrderedDict()
for l in rows:
    for k, v in re.findall(r"(\w+) (\d+)", l[5:]):
        if k.isdigit(): continue
        tot[k] = tot.get(k, 0) + int(v)
n = max(1, len(rows))
# per-sample totals: a launch lasts as long as its slowest wave
import statistics
tots = sorted(sum(int(v) for k, v in re.findall(r"(\w+) (\d+)", l[5:]) if not k.isdigit()) for l in rows)
if tots:
    q = lambda f: tots[min(len(tots) - 1, int(f * len(tots)))]
    print(f"per-sample total cycles: median {q(0.5)}  q90 {q(0.9)}  q99 {q(0.99)}  max {tots[-1]}")
    heavy = [l for l in rows if sum(int(v) for k, v in re.findall(r"(\w+) (\d+)", l[5:]) if not k.isdigit()) >= q(0.99)]
    for l in heavy[:3]: print("  heavy sample:", l)
s = sum(tot.values())
print(f"{len(rows)} samples; cycles per step per wave: {s / n:.0f}")
for k, v in tot.items():
    print(f"  {k:10s} {v / n:10.0f}  {100 * v / s:5.1f} %")
