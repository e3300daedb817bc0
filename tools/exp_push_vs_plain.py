#!/usr/bin/env python3
"""GPU box helper: is the one-block Ant kernel (NB = 1) less accurate than the plain one (NB = 0) on the SAME robot states?
Robot states from an AntUMaze oracle rollout (free space around the start cell in both mazes), stepped with the same actions by
AntUMaze-v0 (NB = 0, 16 lanes) and AntPush-v0 (NB = 1, 32 lanes; block at its spawn position), each against its own oracle model."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
oracle = oracle_lib.load()
n = 2048
plain = mm.make("AntUMaze-v0", num_envs=n, force_vec=True)
push = mm.make("AntPush-v0", num_envs=n, force_vec=True)
opts = [kv.split("=") for kv in sys.argv[1:]]
for k, v in opts:
    plain.set_option(k, float(v)); push.set_option(k, float(v))
rng = np.random.default_rng(3)
st, _ = oracle.reset(plain.model, n, 3)
np.set_printoptions(precision=2, linewidth=200, suppress=True)
f32 = lambda a: a.astype(np.float32).astype(np.float64)
for k in range(61):
    act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
    if k in (3, 6, 10, 20, 40, 60):
        keep = (np.abs(st["qpos"][:, 0]) < 1.5) & (np.abs(st["qpos"][:, 1]) < 1.5)
        res = {}
        for name, env in (("plain", plain), ("push", push)):
            s = {"qpos": f32(st["qpos"]), "qvel": f32(st["qvel"]), "warm": f32(st["warm"]), "t": st["t"].copy()}
            if name == "push":
                z2 = np.zeros((n, 2))
                s = {"qpos": np.concatenate([s["qpos"], z2], 1), "qvel": np.concatenate([s["qvel"], z2], 1), "warm": np.concatenate([s["warm"], z2], 1), "t": s["t"]}
            env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
            env.step(torch.as_tensor(act, device=env.device))
            qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
            oracle.step(env.model, s, act.astype(np.float64), nthreads=16)
            ev = (np.abs(qvel - s["qvel"]) / (1 + np.abs(s["qvel"])))[:, :14]
            res[name] = (ev, s["qvel"][:, :14].copy(), qvel[:, :14].copy())
        # the two oracles must agree on the robot (same physics where nothing touches the block)
        same = np.abs(res["plain"][1] - res["push"][1]).max(1) < 1e-9
        m = keep & same
        for name in ("plain", "push"):
            e = res[name][0][m]
            print(f"step {k:3d} {name:5s} ({m.sum()} envs, oracles agree on {same.sum()}): per-env max  median {np.median(e.max(1)):.1e} 99% {np.quantile(e.max(1), .99):.1e} 99.9% {np.quantile(e.max(1), .999):.1e} max {e.max():.1e} | 99.9% per dof (1e-6): {np.quantile(e, .999, axis=0) * 1e6}")
        d = np.abs(res["plain"][2] - res["push"][2])[m]
        print(f"          device plain vs device push: max |dqvel| {d.max():.1e}, 99.9% {np.quantile(d.max(1), .999):.1e}")
    oracle.step(plain.model, st, act.astype(np.float64), nthreads=16)
