#!/usr/bin/env python3
"""GPU box helper: long auto-reset rollouts with random actions; reports status bits seen and non-finite observations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mujoco_maze_amd as mm
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None
# (AntUMaze at 8192 envs: the two-waves-per-SIMD kernel)
for env_id, n in (("AntUMaze-v0", 4096), ("AntUMaze-v0", 8192), ("Ant4Rooms-v0", 4096), ("AntPush-v0", 2048), ("AntMultiPush-v0", 1024), ("AntPushMaze-v0", 1024), ("PointUMaze-v0", 4096),
                  ("PointPush-v0", 4096), ("PointBilliard-v0", 4096), ("SwimmerUMaze-v0", 4096), ("ReacherUMaze-v0", 4096),
                  ("AntFall-v0", 2048), ("AntMultiFall-v0", 1024), ("PointFall-v0", 4096), ("AntSmallBilliard-v0", 2048)):
    if ONLY and env_id not in ONLY: continue
    env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
    if os.environ.get("MZ_SOAK_RTOL") and env_id.startswith("Ant"): env.set_option("solver_rtol", float(os.environ["MZ_SOAK_RTOL"]))
    seed = int(os.environ.get("MZ_SOAK_SEED", "3"))  # reset seed; the action stream is seeded with seed - 3 (default run: 3 / 0)
    env.reset(seed=seed)
    g = torch.Generator(device=env.device).manual_seed(seed - 3)
    lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
    acts = [lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g) for _ in range(64)]
    k = steps if not (env_id.startswith("AntPushMaze") or env_id.startswith("AntMultiPush") or "Fall" in env_id) else steps // 4
    bits = torch.zeros(n, dtype=torch.int32, device=env.device); nonfinite = 0; dones = 0
    t0 = time.perf_counter()
    for i in range(k):
        obs, rew, done, info = env.step(acts[i % 64])
        if i % 500 == 499:
            bits |= env.status(); nonfinite += int((~torch.isfinite(obs)).sum().item()); dones += int((done != 0).sum().item())
    bits |= env.status()
    torch.cuda.synchronize()
    b = bits.cpu().numpy()
    print(f"{env_id:18s} n={n} steps={k}: {time.perf_counter()-t0:5.1f} s  envs ever flagged: nan {int((b&1!=0).sum())} overflow {int((b&2!=0).sum())} "
          f"maxiter {int((b&4!=0).sum())} collinear {int((b&8!=0).sum())}; non-finite obs samples {nonfinite}; dones sampled {dones}")
    env.close()
