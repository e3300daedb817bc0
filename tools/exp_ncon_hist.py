#!/usr/bin/env python3
"""GPU box helper: how many contacts do the envs of a settled random-action rollout hold?  (first forward evaluation of a step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntPush-v0"; n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True); env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, 8), device=env.device, generator=g) * 60 - 30) for _ in range(16)]
h = np.zeros(64, np.int64); zs = []
for i in range(400):
    env.step(acts[i % 16])
    if i >= 150 and i % 10 == 0:
        _, c = env.debug_forward(acts[(i + 1) % 16])
        h += np.bincount(np.minimum(c.cpu().numpy()[:, 0], 63), minlength=64)
        zs.append(env.get_state()[0][:, 2].cpu().numpy())
tot = h.sum()
print(env_id, "contacts per env (MuJoCo's count):", {k: f"{100 * v / tot:.2f}%" for k, v in enumerate(h) if v})
z = np.concatenate(zs); print("torso height: q01 %.2f q10 %.2f median %.2f" % (np.quantile(z, .01), np.quantile(z, .1), np.median(z)))
