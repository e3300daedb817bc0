#!/usr/bin/env python3
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
"""GPU box helper: throughput of the other configs (not the headline bench)."""
import time, torch
import mujoco_maze_amd as mm
for env_id, n, nu, lo, hi in (("AntPush-v0", 2048, 8, -30, 30), ("Ant4Rooms-v0", 4096, 8, -30, 30), ("AntPushMaze-v0", 2048, 8, -30, 30),
                              ("PointUMaze-v0", 4096, 2, -1, 1), ("Point4Rooms-v0", 4096, 2, -1, 1), ("PointPush-v0", 4096, 2, -1, 1),
                              ("PointPushMaze-v0", 4096, 2, -1, 1), ("SwimmerUMaze-v0", 4096, 2, -1, 1), ("ReacherUMaze-v0", 4096, 1, -1, 1)):
    env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
    env.reset(seed=1)
    g = torch.Generator(device=env.device).manual_seed(0)
    acts = [(torch.rand((n, nu), device=env.device, generator=g) * (hi - lo) + lo) for _ in range(16)]
    if env_id.startswith("Point"):
        for a in acts: a[:, 1] *= 0.25
    for i in range(50): env.step(acts[i % 16])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    K = 200
    for i in range(K): env.step(acts[i % 16])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = env.status().cpu().numpy()
    print(f"{env_id:14s} n={n}  {n*K/dt/1e6:.2f} M env-steps/s  ({dt/K*1e3:.3f} ms/step)  status bits: nan {int((st&1).sum())} overflow {int((st&2).sum())} maxiter {int((st&4).sum())}")
    env.close()
