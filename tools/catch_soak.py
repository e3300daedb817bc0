#!/usr/bin/env python3
"""GPU box helper (debug): replay tools/soak.py's exact action sequence for one config until a status bit appears, then locate the
step and env (snapshots every 100 steps, single-stepping inside the interval) and dump the state / action it was stepped from
to gpurun_out/nan_cases.npy (the format tools/replay_maxiter.py and tools/replay_gpu.py read).
    catch_soak.py <env id> <envs> <steps> <status bit>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
env_id, n, steps, bit = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
env = mm.make(env_id, num_envs=n, auto_reset=True, force_vec=True)
env.reset(seed=3)
g = torch.Generator(device=env.device).manual_seed(0)
lo = torch.as_tensor(env.action_space.low, device=env.device); hi = torch.as_tensor(env.action_space.high, device=env.device)
acts = [lo + (hi - lo) * torch.rand((n, env.nu), device=env.device, generator=g) for _ in range(64)]
snap, snap_i = [x.clone() for x in env.get_state()], 0
found = []
i = 0
while i < steps and not found:
    env.step(acts[i % 64]); i += 1
    if i % 100 == 0:
        if int(((env.status() & bit) != 0).sum()):
            # auto-reset state (episode counters) is part of the record only through t: intervals with a reset inside replay approximately
            env.set_state(*[x.clone() for x in snap])
            base = env.status().clone()
            for j in range(snap_i, i):
                prev = [x.clone() for x in env.get_state()]
                env.step(acts[j % 64])
                st = env.status()
                bad = torch.nonzero(((st & bit) != 0) & ((base & bit) == 0)).flatten()
                if len(bad):
                    for e in bad.tolist()[:4]:
                        found.append(dict(step=j, env=e, qpos=prev[0][e].cpu().numpy(), qvel=prev[1][e].cpu().numpy(), warm=prev[2][e].cpu().numpy(), t=int(prev[3][e]),
                                          act=acts[j % 64][e].cpu().numpy(), after_qpos=env.get_state()[0][e].cpu().numpy()))
                    break
            if not found: print("interval", snap_i, i, "did not reproduce under single-stepping (status was already set before, or a reset inside)")
            break
        snap, snap_i = [x.clone() for x in env.get_state()], i
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/nan_cases.npy", np.array(found, dtype=object), allow_pickle=True)
print("cases", len(found), [(f["step"], f["env"]) for f in found])
