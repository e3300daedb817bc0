#!/usr/bin/env python3
"""GPU box helper: share of envs beyond 1e-5 per checkpoint of the Fall-family parity test (the block's expulsion steps vs later)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mujoco_maze_amd as mm
from tests import oracle_lib
oracle = oracle_lib.load()
for env_id in ("AntFall-v0", "AntMultiFall-v2", "AntMultiFall-v0"):
    n = 1024
    env = mm.make(env_id, num_envs=n)
    cm = env.model
    st, _ = oracle.reset(cm, n, 1)
    rng = np.random.default_rng(0)
    for k in range(41):
        act = rng.uniform(-30, 30, (n, 8)).astype(np.float32)
        if k in (0, 1, 2, 3, 5, 10, 20, 40):
            s = {kk: (v.astype(np.float32).astype(np.float64) if v.dtype != np.int32 else v.copy()) for kk, v in st.items()}
            env.set_state(s["qpos"], s["qvel"], s["warm"], s["t"])
            env.step(torch.as_tensor(act, device=env.device))
            qpos, qvel, _, _ = [x.cpu().numpy() for x in env.get_state()]
            oracle.step(cm, s, act.astype(np.float64), nthreads=16)
            ev = np.abs(qvel - s["qvel"]) - 1e-5 * np.abs(s["qvel"]); ep = np.abs(qpos - s["qpos"]) - 1e-5 * np.abs(s["qpos"])
            e = np.maximum(ev.max(1), ep.max(1))
            print(f"{env_id} k={k:2d}: > 1e-5: {(e > 1e-5).sum():4d}  > 2e-5: {(e > 2e-5).sum():4d}  > 4e-5: {(e > 4e-5).sum():4d}  > 1e-4: {(e > 1e-4).sum():4d}  max {e.max():.1e}")
        oracle.step(cm, st, act.astype(np.float64), nthreads=16)
    env.close()
