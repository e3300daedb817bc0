#!/usr/bin/env python3
"""GPU box helper: PointUMaze-v0 (BASELINE configs[1]) throughput for a lanes-per-env setting: bench_point.py [lanes] [env_id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mujoco_maze_amd as mm
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 0
env_id = sys.argv[2] if len(sys.argv) > 2 else "PointUMaze-v0"
n = 4096
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
if lanes:
    env.set_option("lanes_per_env", lanes)
env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
acts = [(torch.rand((n, 2), device=env.device, generator=g) * 2 - 1) for _ in range(16)]
for a in acts:
    a[:, 1] *= 0.25
for i in range(100):
    env.step(acts[i % 16])
K = 1000
env.set_option("time_kernels", K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 16])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = env.status().cpu().numpy()
print(f"{env_id} lanes={lanes or 'default'} n={n}: {n*K/dt/1e6:.1f} M env-steps/s, {dt/K*1e3:.4f} ms/step, kernel {env.kernel_ms():.4f} ms, bad {int((st & 7 != 0).sum())}")
