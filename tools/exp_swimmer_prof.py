#!/usr/bin/env python3
"""GPU box helper: phase timers of the chain kernel (experiment library csrc/exp_SWPROF.so, planar_kernels.hip built -DMZ_EXP_SWPROF): mean shader
cycles per env.step and phase over the printed workgroups of the last 200 steps."""
import os, re, subprocess, sys, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
env = dict(os.environ, MZ_DEBUG="1", MZ_LIBMAZESTEP_EXPERIMENT=os.path.join(root, "mujoco_maze_amd/csrc/exp_SWPROF.so"))
code = """
import torch, mujoco_maze_amd as mm
n=4096
env = mm.make('%s', num_envs=n, auto_reset=True, force_vec=True); env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
nu = env.nu
acts = [(torch.rand((n, nu), device=env.device, generator=g) * 2 - 1) for _ in range(16)]
for i in range(300): env.step(acts[i %% 16])
torch.cuda.synchronize()
""" % (sys.argv[1] if len(sys.argv) > 1 else "SwimmerUMaze-v0")
out = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True).stdout
rows = [l for l in out.splitlines() if l.startswith("PROF")]
rows = rows[len(rows) // 3:]
tot = collections.OrderedDict()
for l in rows:
    for k, v in re.findall(r"(\w+) (\d+)", l[5:]):
        if k.isdigit(): continue
        tot[k] = tot.get(k, 0) + int(v)
n = max(1, len(rows))
s = sum(tot.values())
print(f"{len(rows)} samples; cycles per step per wave: {s / n:.0f}")
for k, v in tot.items():
    print(f"  {k:10s} {v / n:10.0f}  {100 * v / s:5.1f} %")
