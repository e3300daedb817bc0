#!/usr/bin/env python3
"""GPU box helper: throughput of one env id: bench_env.py ENV_ID [num_envs] [lanes_per_env]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mujoco_maze_amd as mm
env_id = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
if lanes:
    env.set_option("lanes_per_env", lanes)
env.reset(seed=1)
g = torch.Generator(device=env.device).manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device=env.device)
hi = torch.as_tensor(env.action_space.high, device=env.device)
acts = [lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g) for _ in range(16)]
for i in range(100):
    env.step(acts[i % 16])
K = 300
env.set_option("time_kernels", K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(K):
    env.step(acts[i % 16])
torch.cuda.synchronize(); dt = time.perf_counter() - t0
st = env.status().cpu().numpy()
print(f"{env_id} n={n} lanes={lanes or 'default'}: {n*K/dt/1e6:.2f} M env-steps/s, {dt/K*1e3:.3f} ms/step, kernel {env.kernel_ms():.3f} ms, "
      f"status nan {int((st&1).sum())} overflow {int((st&2).sum())} maxiter {int((st&4).sum())}")
