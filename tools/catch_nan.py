#!/usr/bin/env python3
"""GPU box helper (debug): roll an env until some env is flagged (status bit of argv[4], default BAD_STATE); dump the state and action it was stepped from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id = sys.argv[1] if len(sys.argv) > 1 else "AntPushMaze-v0"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
bit = int(sys.argv[4]) if len(sys.argv) > 4 else 1  # status bit to catch: 1 BAD_STATE, 2 CONTACT_OVERFLOW, 4 SOLVER_MAXITER
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
env.reset(seed=3)
g = torch.Generator(device=env.device).manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
found = []
seen = torch.zeros(n, dtype=torch.int32, device=env.device)
for i in range(steps):
    a = lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g)
    prev = [x.clone() for x in env.get_state()]
    env.step(a)
    st = env.status()
    bad = torch.nonzero(((st & bit) != 0) & ((seen & bit) == 0)).flatten()
    seen = st.clone()
    if len(bad):
        for e in bad.tolist()[:4]:
            found.append(dict(step=i, env=e, qpos=prev[0][e].cpu().numpy(), qvel=prev[1][e].cpu().numpy(), warm=prev[2][e].cpu().numpy(), t=int(prev[3][e]),
                              act=a[e].cpu().numpy(), after_qpos=env.get_state()[0][e].cpu().numpy()))
        if len(found) >= 6: break
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/nan_cases.npy", np.array(found, dtype=object), allow_pickle=True)
print("cases", len(found), [(f["step"], f["env"]) for f in found])
